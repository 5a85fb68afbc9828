/* TEST INFRASTRUCTURE — CPU oracle (plain C, float32) of the reference's per-step vehicle update.
 * Not shipped, not a fallback: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline call it.
 *
 * Restates (all paths relative to /root/reference):
 *   physics::FreeCar::Throttle/Brake/Turn/Step, DampenSpeed   nocturne/cpp/src/physics/FreeCar.cpp:66-186
 *   constants                                                   nocturne/cpp/include/physics/defines.h:4-11
 *   b2Body::SetLinearVelocity/SetAngularVelocity/SetAwake       third_party/box2d/include/box2d/b2_body.h
 *   b2Island::Solve integrate / clamp / sleep (contact-free)    third_party/box2d/src/dynamics/b2_island.cpp:194-229,279-310,349-392
 *   patched b2_maxTranslation = 5.0f                            third_party/box2d/include/box2d/b2_common.h:95
 *   Vehicle::CreatePhysicsBody / setters / Step readback        nocturne/cpp/src/vehicle.cc:25-55,75-179
 *   Object::BoundingPolygon, Object::Velocity                   nocturne/cpp/src/object.cc:14-28, include/object.h:152-154
 *   Scenario::Step / UpdateCollision                            nocturne/cpp/src/scenario.cc:266-328
 *   ConvexPolygon::Intersects / Separates, Polygon::GetAABB     nocturne/cpp/src/geometry/polygon.cc:19-44,84-98
 *   Intersects(ConvexPolygon, LineSegment)                      nocturne/cpp/src/geometry/intersection.cc:200-232
 *   AABB::Intersects (strict)                                   nocturne/cpp/include/geometry/aabb.h:47-50
 *
 * Tier: CONTACT-FREE.  Box2D's contact solver (cars pushing each other apart when their boxes overlap) is
 * not restated; rollouts in which two live boxes overlap diverge from the reference after the overlap
 * (DESIGN.md "scope").  Collision FLAGS are exact in every case.
 *
 * Pinned against oracle/_ref/libref_sim.so (the real FreeCar + Box2D + geometry sources) by
 * tests/test_sim_oracle.py and the fixtures tests/golden/physics_*.npz.
 * Build: make -C oracle oracle   (-O2 -ffp-contract=off: no FMA contraction).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXSPEED 50.f
#define MAXREVERSESPEED -5.f
#define MAXTHROTTLEACCEL 1.0f
#define MAXTHROTTLEREVERSEACCEL 0.f
#define MAXBRAKEACCEL 1.0f
#define SIDESPEEDDAMPING 25.f
#define ANGULARDAMPING 10.f
#define B2_PI 3.14159265359f
#define B2_MAXTRANSLATION 5.0f
#define B2_MAXROTATION (0.5f * B2_PI)
#define B2_LINSLEEPTOL 0.01f
#define B2_ANGSLEEPTOL (2.0f / 180.0f * B2_PI)
#define B2_TIMETOSLEEP 0.5f

typedef struct {
  float length, width;
  /* Object state (what Python reads) */
  float px, py, heading, speed;
  /* Box2D body: sweep.c (centre of mass), sweep.a, velocities; xf.p is (px,py) above; lc = sweep.localCenter */
  float cx, cy, a, vx, vy, w, sleep_time, lcx, lcy;
  int awake;
  /* FreeCar controls */
  float throttle, brake, steer;
  unsigned char coll_veh, coll_edge;
} Veh;

typedef struct {
  int n, n_seg;
  Veh* v;
  float* segs;
} Sim;

/* b2PolygonShape::SetAsBox(hx,hy) + ComputeMass(density 20) + b2Body::ResetMassData: the body's local centre
 * of mass.  Mathematically (0,0); in float32 the triangle-fan sum leaves a ~1e-8 residue that shifts the body
 * origin by an ulp now and then, so it has to be carried.  third_party/box2d/src/collision/b2_polygon_shape.cpp:36-48,
 * 357-431; src/dynamics/b2_body.cpp ResetMassData; FreeCar.cpp:34-40. */
static void local_center(float width, float length, float* lcx, float* lcy) {
  float hx = width / 2, hy = length / 2;
  float vx[4] = {-hx, hx, hx, -hx}, vy[4] = {-hy, -hy, hy, hy};
  float cx = 0.0f, cy = 0.0f, area = 0.0f;
  float sx = vx[0], sy = vy[0];
  const float k_inv3 = 1.0f / 3.0f;
  for (int i = 0; i < 4; ++i) {
    float e1x = vx[i] - sx, e1y = vy[i] - sy;
    float e2x = (i + 1 < 4 ? vx[i + 1] : vx[0]) - sx, e2y = (i + 1 < 4 ? vy[i + 1] : vy[0]) - sy;
    float D = e1x * e2y - e1y * e2x;
    float ta = 0.5f * D;
    area += ta;
    float k = ta * k_inv3;
    cx += k * (e1x + e2x);
    cy += k * (e1y + e2y);
  }
  float mass = 20.f * area;
  float inv_area = 1.0f / area;
  cx *= inv_area; cy *= inv_area;
  float mcx = cx + sx, mcy = cy + sy;           /* massData->center */
  float lx = mass * mcx, ly = mass * mcy;       /* localCenter += massData.mass * massData.center */
  float inv_mass = 1.0f / mass;
  *lcx = lx * inv_mass; *lcy = ly * inv_mass;
}

/* b2Body::SetTransform(position, angle): sweep.c = b2Mul(xf, localCenter) */
static void set_transform(Veh* v, float x, float y, float angle) {
  float qs = sinf(angle), qc = cosf(angle);
  v->px = x; v->py = y; v->a = angle;
  v->cx = (qc * v->lcx - qs * v->lcy) + x;
  v->cy = (qs * v->lcx + qc * v->lcy) + y;
}

static float dampen(float speed, float target, float damping, float dt) {
  float red = damping * dt;
  if (speed - target > red) return speed - red;
  if (speed - target < -red) return speed + red;
  return target;
}

static void set_awake_true(Veh* v) { v->awake = 1; v->sleep_time = 0.0f; }

static void freecar_step(Veh* v, float dt) {
  float target = 0.f, acc = 0.f;
  if (v->throttle > 0.f) {
    if (v->throttle > v->brake) { target = MAXSPEED; acc = v->throttle - v->brake; }
    else { target = 0.f; acc = v->brake - v->throttle; }
  } else {
    if (v->throttle < -v->brake) { target = MAXREVERSESPEED; acc = -v->throttle - v->brake; }
    else { target = 0.f; acc = v->brake + v->throttle; }
  }
  float ang = v->w;
  float beta = (float)atan(0.5 * (double)tanf(v->steer));
  float c = cosf(v->a + beta);
  float s = sinf(v->a + beta);
  float fx = -s, fy = c, rx = c, ry = s;
  float sf = v->vx * fx + v->vy * fy;
  float sr = v->vx * rx + v->vy * ry;
  float dv = acc * dt;
  if (sf < target) sf = fminf(sf + dv, target);
  else sf = fmaxf(sf - dv, target);
  float steer_w = 0.f;
  if (fabs((double)v->steer) > 0.0000001) {
    float ray = 1.f / tanf(v->steer) * v->length / cosf(beta);
    steer_w = sf / ray;
  }
  sr = dampen(sr, 0, SIDESPEEDDAMPING, dt);
  ang = dampen(ang, steer_w, ANGULARDAMPING, dt);
  float nvx = rx * sr + fx * sf;
  float nvy = ry * sr + fy * sf;
  if (nvx * nvx + nvy * nvy > 0.0f) set_awake_true(v);
  v->vx = nvx; v->vy = nvy;
  if (ang * ang > 0.0f) set_awake_true(v);
  v->w = ang;
}

static void island_solve(Veh* v, float h) {
  if (!v->awake) return;
  float tx = h * v->vx, ty = h * v->vy;
  if (tx * tx + ty * ty > B2_MAXTRANSLATION * B2_MAXTRANSLATION) {
    float ratio = B2_MAXTRANSLATION / sqrtf(tx * tx + ty * ty);
    v->vx *= ratio; v->vy *= ratio;
  }
  float rot = h * v->w;
  if (rot * rot > B2_MAXROTATION * B2_MAXROTATION) {
    float ratio = B2_MAXROTATION / fabsf(rot);
    v->w *= ratio;
  }
  v->cx += h * v->vx; v->cy += h * v->vy;
  v->a += h * v->w;
  if (v->w * v->w > B2_ANGSLEEPTOL * B2_ANGSLEEPTOL ||
      v->vx * v->vx + v->vy * v->vy > B2_LINSLEEPTOL * B2_LINSLEEPTOL) {
    v->sleep_time = 0.0f;
  } else {
    v->sleep_time += h;
  }
  {                                             /* b2Body::SynchronizeTransform (b2_island.cpp copy-back) */
    float qs = sinf(v->a), qc = cosf(v->a);
    v->px = v->cx - (qc * v->lcx - qs * v->lcy);
    v->py = v->cy - (qs * v->lcx + qc * v->lcy);
  }
  if (v->sleep_time >= B2_TIMETOSLEEP) {  /* single-body island, positionSolved is true without contacts */
    v->awake = 0; v->sleep_time = 0.0f; v->vx = v->vy = 0.0f; v->w = 0.0f;
  }
}

static void corners(const Veh* v, float* p /*[8]*/) {
  float st = sinf(v->heading), ct = cosf(v->heading);
  float hx[4] = {v->length * 0.5f, -v->length * 0.5f, -v->length * 0.5f, v->length * 0.5f};
  float hy[4] = {v->width * 0.5f, v->width * 0.5f, -v->width * 0.5f, -v->width * 0.5f};
  for (int k = 0; k < 4; ++k) {
    p[2 * k] = (hx[k] * ct - hy[k] * st) + v->px;
    p[2 * k + 1] = (hx[k] * st + hy[k] * ct) + v->py;
  }
}

static float cross2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

static int separates(const float* e0, const float* e1, const float* poly, int n) {
  float dx = e1[0] - e0[0], dy = e1[1] - e0[1];
  for (int k = 0; k < n; ++k) {
    if (cross2(poly[2 * k] - e0[0], poly[2 * k + 1] - e0[1], dx, dy) <= 0.0f) return 0;
  }
  return 1;
}

static int poly_poly(const float* a, int na, const float* b, int nb) {
  for (int k = 0; k < na; ++k) {
    const float* e0 = (k == na - 1) ? a + 2 * (na - 1) : a + 2 * k;
    const float* e1 = (k == na - 1) ? a : a + 2 * (k + 1);
    if (separates(e0, e1, b, nb)) return 0;
  }
  for (int k = 0; k < nb; ++k) {
    const float* e0 = (k == nb - 1) ? b + 2 * (nb - 1) : b + 2 * k;
    const float* e1 = (k == nb - 1) ? b : b + 2 * (k + 1);
    if (separates(e0, e1, a, na)) return 0;
  }
  return 1;
}

static int poly_contains(const float* a, int n, float x, float y) {
  for (int i = 1; i < n; ++i) {
    if (cross2(x - a[2 * (i - 1)], y - a[2 * (i - 1) + 1], a[2 * i] - a[2 * (i - 1)], a[2 * i + 1] - a[2 * (i - 1) + 1]) > 0.0f)
      return 0;
  }
  return cross2(x - a[2 * (n - 1)], y - a[2 * (n - 1) + 1], a[0] - a[2 * (n - 1)], a[1] - a[2 * (n - 1) + 1]) <= 0.0f;
}

static int poly_seg(const float* a, int n, const float* s) {
  if (s[0] == s[2] && s[1] == s[3]) return poly_contains(a, n, s[0], s[1]);
  float dx = s[2] - s[0], dy = s[3] - s[1];
  float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
  for (int k = 0; k < n; ++k) {
    float cur = cross2(a[2 * k] - s[0], a[2 * k + 1] - s[1], dx, dy);
    mn = fminf(mn, cur); mx = fmaxf(mx, cur);
  }
  if (mx < 0.0f || mn > 0.0f) return 0;
  for (int k = 0; k < n; ++k) {
    const float* e0 = (k == n - 1) ? a + 2 * (n - 1) : a + 2 * k;
    const float* e1 = (k == n - 1) ? a : a + 2 * (k + 1);
    float cx = e1[0] - e0[0], cy = e1[1] - e0[1];
    float v0 = cross2(s[0] - e0[0], s[1] - e0[1], cx, cy);
    float v1 = cross2(s[2] - e0[0], s[3] - e0[1], cx, cy);
    if (v0 > 0.0f && v1 > 0.0f) return 0;
  }
  return 1;
}

static void aabb_of(const float* p, int n, float* bb) {
  bb[0] = bb[1] = 3.402823466e+38f; bb[2] = bb[3] = -3.402823466e+38f;
  for (int k = 0; k < n; ++k) {
    bb[0] = fminf(bb[0], p[2 * k]); bb[2] = fmaxf(bb[2], p[2 * k]);
    bb[1] = fminf(bb[1], p[2 * k + 1]); bb[3] = fmaxf(bb[3], p[2 * k + 1]);
  }
}

static int aabb_hit(const float* a, const float* b) {
  return a[0] < b[2] && a[2] > b[0] && a[1] < b[3] && a[3] > b[1];
}

static void update_collision(Sim* s) {
  int n = s->n;
  float* P = (float*)malloc(sizeof(float) * 8 * (n > 0 ? n : 1));
  float* B = (float*)malloc(sizeof(float) * 4 * (n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) { corners(&s->v[i], P + 8 * i); aabb_of(P + 8 * i, 4, B + 4 * i); }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      if (i == j || !aabb_hit(B + 4 * i, B + 4 * j)) continue;
      if (poly_poly(P + 8 * i, 4, P + 8 * j, 4)) s->v[i].coll_veh = 1;
    }
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < s->n_seg; ++k) {
      const float* sg = s->segs + 4 * k;
      float sb[4] = {fminf(sg[0], sg[2]), fminf(sg[1], sg[3]), fmaxf(sg[0], sg[2]), fmaxf(sg[1], sg[3])};
      if (!aabb_hit(B + 4 * i, sb)) continue;
      if (poly_seg(P + 8 * i, 4, sg)) s->v[i].coll_edge = 1;
    }
  free(P); free(B);
}

void* orasim_create(int n, const float* length, const float* width, const float* x, const float* y,
                    const float* heading, const float* speed, int n_seg, const float* segs) {
  Sim* s = (Sim*)calloc(1, sizeof(Sim));
  s->n = n; s->n_seg = n_seg;
  s->v = (Veh*)calloc(n > 0 ? n : 1, sizeof(Veh));
  s->segs = (float*)malloc(sizeof(float) * 4 * (n_seg > 0 ? n_seg : 1));
  if (n_seg > 0) memcpy(s->segs, segs, sizeof(float) * 4 * n_seg);
  for (int i = 0; i < n; ++i) {
    Veh* v = &s->v[i];
    v->length = length[i]; v->width = width[i];
    v->px = x[i]; v->py = y[i]; v->heading = heading[i]; v->speed = speed[i];
    local_center(v->width, v->length, &v->lcx, &v->lcy);
    set_transform(v, 0.f, 0.f, (float)((double)v->heading - M_PI * 0.5f));   /* SetAngle, vehicle.cc:168 */
    set_transform(v, x[i], y[i], v->a);                                      /* SetPosition, vehicle.cc:169 */
    float c = cosf(v->heading), sn = sinf(v->heading);
    v->vx = v->speed * c; v->vy = v->speed * sn;        /* BaseCar::SetSpeed: plain assignment + wake */
    v->w = 0.f; v->sleep_time = 0.f; v->awake = 1;
    v->throttle = v->brake = v->steer = 0.f;
  }
  update_collision(s);
  return s;
}

void orasim_set_action(void* h, int i, double accel, double steer) {
  Veh* v = &((Sim*)h)->v[i];
  if (accel > 0.0) {                       /* Throttle(value>0), FreeCar.cpp:66-73 */
    float a = (float)accel;
    v->throttle = (a > 0) ? MAXTHROTTLEACCEL * a : MAXTHROTTLEREVERSEACCEL * a;
    v->brake = 0.f;
  } else {                                 /* Brake, FreeCar.cpp:75-81 */
    float b = (float)fabs(accel);
    if (!(fabsf(b) < 0.001)) { v->throttle = 0; v->brake = MAXBRAKEACCEL * b; }
  }
  v->steer = (float)steer;
}

void orasim_set_position(void* h, int i, float x, float y) {
  Veh* v = &((Sim*)h)->v[i];
  set_transform(v, x, y, v->a);                 /* b2Body::SetTransform keeps the angle, does not wake */
}

void orasim_step(void* h, float dt) {
  Sim* s = (Sim*)h;
  for (int i = 0; i < s->n; ++i) freecar_step(&s->v[i], dt);
  for (int i = 0; i < s->n; ++i) island_solve(&s->v[i], dt);
  for (int i = 0; i < s->n; ++i) {
    Veh* v = &s->v[i];
    v->coll_veh = v->coll_edge = 0;             /* position_ <- m_xf.p (already in px,py) */
    v->speed = sqrtf(v->vx * v->vx + v->vy * v->vy);
    v->heading = (float)((double)v->a + M_PI * 0.5f);
  }
  update_collision(s);
}

void orasim_get_state(void* h, float* out, unsigned char* cv, unsigned char* ce) {
  Sim* s = (Sim*)h;
  for (int i = 0; i < s->n; ++i) {
    const Veh* v = &s->v[i];
    out[6 * i + 0] = v->px; out[6 * i + 1] = v->py; out[6 * i + 2] = v->heading; out[6 * i + 3] = v->speed;
    out[6 * i + 4] = v->speed * cosf(v->heading);
    out[6 * i + 5] = v->speed * sinf(v->heading);
    cv[i] = v->coll_veh; ce[i] = v->coll_edge;
  }
}

void orasim_get_body(void* h, float* out) {
  Sim* s = (Sim*)h;
  for (int i = 0; i < s->n; ++i) {
    const Veh* v = &s->v[i];
    out[6 * i + 0] = v->px; out[6 * i + 1] = v->py; out[6 * i + 2] = v->a;
    out[6 * i + 3] = v->vx; out[6 * i + 4] = v->vy; out[6 * i + 5] = v->w;
  }
}

void orasim_destroy(void* h) {
  Sim* s = (Sim*)h;
  free(s->v); free(s->segs); free(s);
}

int orageo_poly_poly(const float* a, int na, const float* b, int nb) { return poly_poly(a, na, b, nb); }
int orageo_poly_seg(const float* a, int na, const float* seg) { return poly_seg(a, na, seg); }

/* Object::KinematicBicycleStep (nocturne/cpp/src/object.cc:126-137), optional integrator mode.
 * state = x, y, heading, speed (in/out). max_speed is float max in the reference (object.h:189). */
void orasim_kinematic_step(float* st, float length, float accel, float steer, float dt) {
  const float kPi = 3.14159265358979323846f, kTwoPi = 2.0f * 3.14159265358979323846f;
  float v = st[3] + 0.5f * accel * dt;
  float tan_delta = tanf(steer);
  float beta = atanf(0.5f * tan_delta);
  float dx = v * cosf(st[2] + beta), dy = v * sinf(st[2] + beta);
  float w = v * cosf(beta) * tan_delta / length;
  st[0] += dx * dt; st[1] += dy * dt;
  float ang = fmodf(st[2] + w * dt, kTwoPi);
  st[2] = ang > kPi ? ang - kTwoPi : (ang < -kPi ? ang + kTwoPi : ang);
  st[3] = st[3] + accel * dt;
}
