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
 *   b2CollidePolygons, b2ClipSegmentToLine, b2WorldManifold    third_party/box2d/src/collision/b2_collide_polygon.cpp:26-244, b2_collision.cpp:26-90,205-237
 *   b2Contact::Update (manifold, impulse matching, wake)        third_party/box2d/src/dynamics/b2_contact.cpp:165-245
 *   b2World::Solve (island DFS), b2Island::Solve                third_party/box2d/src/dynamics/b2_world.cpp:393-560, b2_island.cpp:194-392
 *   b2ContactSolver (init, warm start, velocity / block solver, position)   third_party/box2d/src/dynamics/b2_contact_solver.cpp:52-760
 *   b2PolygonShape::ComputeMass / b2Body::ResetMassData         b2_polygon_shape.cpp:357-431, b2_body.cpp:290-354
 *
 * Tier: CONTACTS INCLUDED.  Box-box contacts between vehicles (friction 0.2, restitution 0, density 20, polygon skin
 * 0.01, 8 velocity / 3 position iterations, warm starting, block solver) are restated and are bit-exact against the
 * real Box2D: two-body islands, pile-ups, and lots of up to 64 cars that overlap from the start (live check in
 * tests/test_oracle_pinned.py).  Gauss-Seidel is order dependent, so the broad phase is restated as well, down to
 * b2DynamicTree (its shape fixes the order in which contacts that begin in the same step are created).
 * TOI sub-stepping never triggers between two non-bullet dynamic bodies
 * (b2_world.cpp SolveTOI).  Collision FLAGS are exact in every case.
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
  float cx, cy, a, vx, vy, w, sleep_time, lcx, lcy, inv_mass, inv_i;
  int awake;
  /* FreeCar controls */
  float throttle, brake, steer;
  unsigned char coll_veh, coll_edge;
  /* Object::expert_control_ for the next step + the logged state of that step (orasim_set_expert) */
  int ex_on;
  float ex_x, ex_y, ex_heading, ex_speed;
} Veh;

/* b2Manifold of one vehicle pair (i < j: fixture A = vehicle i, B = vehicle j) with the accumulated impulses */
typedef struct { float lx, ly, ni, ti; unsigned id; } MPoint;
typedef struct { MPoint p[2]; float lnx, lny, lpx, lpy; int type, count, touching; } Manifold;
enum { M_FACE_A = 1, M_FACE_B = 2 };

struct TNode { float bb[4]; int parent, c1, c2, height; };   /* b2TreeNode; `parent` doubles as `next` in the free list */
typedef struct {
  int n, n_seg;
  Veh* v;
  float* segs;
  Manifold* man;          /* [n * n], entry i * n + j for i < j */
  float inv_dt0;          /* b2World::m_inv_dt0 */
  /* broad-phase bookkeeping, kept only to reproduce the ORDER in which Box2D creates contacts (it decides the
   * Gauss-Seidel order inside islands of three or more bodies): fat AABB per proxy (b2DynamicTree::MoveProxy),
   * move buffer (b2BroadPhase::BufferMove / UpdatePairs), contact existence + creation stamp per pair */
  float* fat;             /* [n][4] lower x, y, upper x, y */
  char* moved;            /* [n] b2TreeNode::moved */
  int* move_buf; int n_move, cap_move;
  char* c_exists;         /* [n * n] */
  int* c_stamp;           /* [n * n] creation order (larger = newer) */
  int stamp, new_contacts;
  float* sweep0;          /* [n][3] sweep.c0, sweep.a0 of the last island solve */
  /* b2DynamicTree: its shape decides the order in which one query reports its hits, hence the order in which contacts that
   * begin in the same step are created.  Leaf of vehicle i = node leaf_of[i] (its box is fat[i]) */
  struct TNode* tn; int t_root, t_free, t_cap;
  int* leaf_of; int* veh_of;
} Sim;

/* b2PolygonShape::SetAsBox(hx,hy) + ComputeMass(density 20) + b2Body::ResetMassData: the body's local centre
 * of mass.  Mathematically (0,0); in float32 the triangle-fan sum leaves a ~1e-8 residue that shifts the body
 * origin by an ulp now and then, so it has to be carried.  third_party/box2d/src/collision/b2_polygon_shape.cpp:36-48,
 * 357-431; src/dynamics/b2_body.cpp ResetMassData; FreeCar.cpp:34-40. */
static void local_center(float width, float length, float* lcx, float* lcy, float* inv_mass, float* inv_i) {
  float hx = width / 2, hy = length / 2;
  float vx[4] = {-hx, hx, hx, -hx}, vy[4] = {-hy, -hy, hy, hy};
  float cx = 0.0f, cy = 0.0f, area = 0.0f, I = 0.0f;
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
    float intx2 = e1x * e1x + e2x * e1x + e2x * e2x;
    float inty2 = e1y * e1y + e2y * e1y + e2y * e2y;
    I += (0.25f * k_inv3 * D) * (intx2 + inty2);
  }
  float mass = 20.f * area;
  float inv_area = 1.0f / area;
  cx *= inv_area; cy *= inv_area;
  float mcx = cx + sx, mcy = cy + sy;           /* massData->center */
  float lx = mass * mcx, ly = mass * mcy;       /* localCenter += massData.mass * massData.center */
  float im = 1.0f / mass;
  *lcx = lx * im; *lcy = ly * im;
  float mI = 20.f * I;                                                   /* massData->I = density * I ... */
  mI += mass * ((mcx * mcx + mcy * mcy) - (cx * cx + cy * cy));         /* ... shifted to the shape origin */
  float bI = mI - mass * (*lcx * *lcx + *lcy * *lcy);                   /* b2Body: about the centre of mass */
  *inv_mass = im; *inv_i = 1.0f / bI;
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

/* ---------------------------------------------------------------------------------------------------- Box2D step
 * Plain-C restatement of what b2World::Step does to a world of dynamic boxes (no gravity, no damping, no joints, no
 * bullets): contact update, islands, contact solver, integration, sleep.  Every expression keeps the operand order of
 * the C++ source (float32, no contraction). */
typedef struct { float x, y; } V2;
typedef struct { float s, c; } Rot;
typedef struct { V2 p; Rot q; } Xf;
#define B2_LINEAR_SLOP 0.005f
#define B2_POLY_RADIUS (2.0f * B2_LINEAR_SLOP)
#define B2_BAUMGARTE 0.2f
#define B2_MAX_LIN_CORR 0.2f
#define B2_EPS 1.1920928955078125e-7f          /* FLT_EPSILON */
#define B2_FLT_MAX 3.402823466e+38F

static float b2maxf(float a, float b) { return a > b ? a : b; }     /* b2Max / b2Min as written in b2_math.h */
static float b2minf(float a, float b) { return a < b ? a : b; }
static V2 v2(float x, float y) { V2 r = {x, y}; return r; }
static V2 rot_mul(Rot q, V2 v) { return v2(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
static V2 rot_mulT(Rot q, V2 v) { return v2(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
static V2 xf_mul(Xf t, V2 v) { return v2((t.q.c * v.x - t.q.s * v.y) + t.p.x, (t.q.s * v.x + t.q.c * v.y) + t.p.y); }
static V2 xf_mulT(Xf t, V2 v) {
  float px = v.x - t.p.x, py = v.y - t.p.y;
  return v2(t.q.c * px + t.q.s * py, -t.q.s * px + t.q.c * py);
}
static Xf xf_mulT_xf(Xf A, Xf B) {             /* b2MulT(A, B) */
  Xf C;
  C.q.s = A.q.c * B.q.s - A.q.s * B.q.c;
  C.q.c = A.q.c * B.q.c + A.q.s * B.q.s;
  C.p = rot_mulT(A.q, v2(B.p.x - A.p.x, B.p.y - A.p.y));
  return C;
}
static float dot2(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
static float crossvv(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
static V2 cross_vs(V2 a, float s) { return v2(s * a.y, -s * a.x); }     /* b2Cross(vec, scalar) */
static V2 cross_sv(float s, V2 a) { return v2(-s * a.y, s * a.x); }     /* b2Cross(scalar, vec) */
static Xf body_xf(const Veh* v) { Xf t; t.p = v2(v->px, v->py); t.q.s = sinf(v->a); t.q.c = cosf(v->a); return t; }

typedef struct { V2 v[4], n[4]; } Box;         /* b2PolygonShape::SetAsBox(width/2, length/2), FreeCar.cpp:39 */
static Box box_of(const Veh* v) {
  float hx = v->width / 2, hy = v->length / 2;
  Box b;
  b.v[0] = v2(-hx, -hy); b.v[1] = v2(hx, -hy); b.v[2] = v2(hx, hy); b.v[3] = v2(-hx, hy);
  b.n[0] = v2(0.0f, -1.0f); b.n[1] = v2(1.0f, 0.0f); b.n[2] = v2(0.0f, 1.0f); b.n[3] = v2(-1.0f, 0.0f);
  return b;
}

typedef struct { V2 v; unsigned char ia, ib, ta, tb; } ClipV;   /* b2ClipVertex: id.cf = indexA, indexB, typeA, typeB */
enum { CF_VERTEX = 0, CF_FACE = 1 };

static float find_max_separation(int* edge, const Box* p1, Xf xf1, const Box* p2, Xf xf2) {
  Xf xf = xf_mulT_xf(xf2, xf1);
  int best = 0;
  float max_sep = -B2_FLT_MAX;
  for (int i = 0; i < 4; ++i) {
    V2 n = rot_mul(xf.q, p1->n[i]);
    V2 v1 = xf_mul(xf, p1->v[i]);
    float si = B2_FLT_MAX;
    for (int j = 0; j < 4; ++j) {
      float sij = dot2(n, v2(p2->v[j].x - v1.x, p2->v[j].y - v1.y));
      if (sij < si) si = sij;
    }
    if (si > max_sep) { max_sep = si; best = i; }
  }
  *edge = best;
  return max_sep;
}

static int clip_segment(ClipV out[2], const ClipV in[2], V2 normal, float offset, int vertex_index_a) {
  int count = 0;
  float d0 = dot2(normal, in[0].v) - offset;
  float d1 = dot2(normal, in[1].v) - offset;
  if (d0 <= 0.0f) out[count++] = in[0];
  if (d1 <= 0.0f) out[count++] = in[1];
  if (d0 * d1 < 0.0f) {
    float interp = d0 / (d0 - d1);
    out[count].v = v2(in[0].v.x + interp * (in[1].v.x - in[0].v.x), in[0].v.y + interp * (in[1].v.y - in[0].v.y));
    out[count].ia = (unsigned char)vertex_index_a;
    out[count].ib = in[0].ib;
    out[count].ta = CF_VERTEX;
    out[count].tb = CF_FACE;
    ++count;
  }
  return count;
}

/* b2CollidePolygons; impulses of the points are left untouched (set by contact_update) */
static void collide_boxes(Manifold* m, const Box* A, Xf xfA, const Box* B, Xf xfB) {
  m->count = 0;
  const float total_radius = B2_POLY_RADIUS + B2_POLY_RADIUS;
  int edgeA = 0, edgeB = 0;
  float sepA = find_max_separation(&edgeA, A, xfA, B, xfB);
  if (sepA > total_radius) return;
  float sepB = find_max_separation(&edgeB, B, xfB, A, xfA);
  if (sepB > total_radius) return;
  const Box *p1, *p2;
  Xf xf1, xf2;
  int edge1, flip;
  const float k_tol = 0.1f * B2_LINEAR_SLOP;
  if (sepB > sepA + k_tol) { p1 = B; p2 = A; xf1 = xfB; xf2 = xfA; edge1 = edgeB; m->type = M_FACE_B; flip = 1; }
  else { p1 = A; p2 = B; xf1 = xfA; xf2 = xfB; edge1 = edgeA; m->type = M_FACE_A; flip = 0; }
  /* b2FindIncidentEdge */
  ClipV inc[2];
  {
    V2 normal1 = rot_mulT(xf2.q, rot_mul(xf1.q, p1->n[edge1]));
    int index = 0;
    float min_dot = B2_FLT_MAX;
    for (int i = 0; i < 4; ++i) {
      float d = dot2(normal1, p2->n[i]);
      if (d < min_dot) { min_dot = d; index = i; }
    }
    int i1 = index, i2 = i1 + 1 < 4 ? i1 + 1 : 0;
    inc[0].v = xf_mul(xf2, p2->v[i1]); inc[0].ia = (unsigned char)edge1; inc[0].ib = (unsigned char)i1; inc[0].ta = CF_FACE; inc[0].tb = CF_VERTEX;
    inc[1].v = xf_mul(xf2, p2->v[i2]); inc[1].ia = (unsigned char)edge1; inc[1].ib = (unsigned char)i2; inc[1].ta = CF_FACE; inc[1].tb = CF_VERTEX;
  }
  int iv1 = edge1, iv2 = edge1 + 1 < 4 ? edge1 + 1 : 0;
  V2 v11 = p1->v[iv1], v12 = p1->v[iv2];
  V2 lt = v2(v12.x - v11.x, v12.y - v11.y);
  {                                               /* b2Vec2::Normalize */
    float len = sqrtf(lt.x * lt.x + lt.y * lt.y);
    if (!(len < B2_EPS)) { float inv = 1.0f / len; lt.x *= inv; lt.y *= inv; }
  }
  V2 local_normal = cross_vs(lt, 1.0f);
  V2 plane_point = v2(0.5f * (v11.x + v12.x), 0.5f * (v11.y + v12.y));
  V2 tangent = rot_mul(xf1.q, lt);
  V2 normal = cross_vs(tangent, 1.0f);
  v11 = xf_mul(xf1, v11);
  v12 = xf_mul(xf1, v12);
  float front_offset = dot2(normal, v11);
  float side1 = -dot2(tangent, v11) + total_radius;
  float side2 = dot2(tangent, v12) + total_radius;
  ClipV c1[2], c2[2];
  int np = clip_segment(c1, inc, v2(-tangent.x, -tangent.y), side1, iv1);
  if (np < 2) return;
  np = clip_segment(c2, c1, tangent, side2, iv2);
  if (np < 2) return;
  m->lnx = local_normal.x; m->lny = local_normal.y;
  m->lpx = plane_point.x; m->lpy = plane_point.y;
  int pc = 0;
  for (int i = 0; i < 2; ++i) {
    float separation = dot2(normal, c2[i].v) - front_offset;
    if (separation <= total_radius) {
      MPoint* cp = &m->p[pc];
      V2 lp = xf_mulT(xf2, c2[i].v);
      cp->lx = lp.x; cp->ly = lp.y;
      unsigned char ia = c2[i].ia, ib = c2[i].ib, ta = c2[i].ta, tb = c2[i].tb;
      if (flip) { unsigned char t; t = ia; ia = ib; ib = t; t = ta; ta = tb; tb = t; }
      cp->id = (unsigned)ia | ((unsigned)ib << 8) | ((unsigned)ta << 16) | ((unsigned)tb << 24);
      ++pc;
    }
  }
  m->count = pc;
}

/* ---- broad phase (third_party/box2d/src/collision/b2_dynamic_tree.cpp:107-195, b2_broad_phase.cpp:60-131,
 *      include/box2d/b2_broad_phase.h:172-216, src/dynamics/b2_fixture.cpp:156-178, b2_body.cpp:447-469) */
#define B2_AABB_EXT 0.1f
#define B2_AABB_MULT 4.0f
static void shape_aabb(const Veh* v, Xf xf, float* bb) {       /* b2PolygonShape::ComputeAABB */
  Box b = box_of(v);
  V2 lo = xf_mul(xf, b.v[0]), hi = lo;
  for (int i = 1; i < 4; ++i) {
    V2 p = xf_mul(xf, b.v[i]);
    lo = v2(b2minf(lo.x, p.x), b2minf(lo.y, p.y));
    hi = v2(b2maxf(hi.x, p.x), b2maxf(hi.y, p.y));
  }
  bb[0] = lo.x - B2_POLY_RADIUS; bb[1] = lo.y - B2_POLY_RADIUS; bb[2] = hi.x + B2_POLY_RADIUS; bb[3] = hi.y + B2_POLY_RADIUS;
}
static void buffer_move(Sim* s, int i) {
  if (s->n_move == s->cap_move) { s->cap_move = s->cap_move ? 2 * s->cap_move : 64; s->move_buf = (int*)realloc(s->move_buf, sizeof(int) * s->cap_move); }
  s->move_buf[s->n_move++] = i;
}
static int aabb_contains(const float* a, const float* b) { return a[0] <= b[0] && a[1] <= b[1] && b[2] <= a[2] && b[3] <= a[3]; }
static int aabb_overlap(const float* a, const float* b) {      /* b2TestOverlap(aabb, aabb) */
  float d1x = b[0] - a[2], d1y = b[1] - a[3], d2x = a[0] - b[2], d2y = a[1] - b[3];
  if (d1x > 0.0f || d1y > 0.0f) return 0;
  if (d2x > 0.0f || d2y > 0.0f) return 0;
  return 1;
}
/* ---- b2DynamicTree (third_party/box2d/src/collision/b2_dynamic_tree.cpp:57-105 node pool, :198-332 InsertLeaf,
 *      :334-393 RemoveLeaf, :397-534 Balance; include/box2d/b2_dynamic_tree.h:187-220 Query, b2_collision.h AABB helpers) */
#define T_NULL (-1)
static void bb_combine(float* o, const float* a, const float* b) {
  o[0] = b2minf(a[0], b[0]); o[1] = b2minf(a[1], b[1]); o[2] = b2maxf(a[2], b[2]); o[3] = b2maxf(a[3], b[3]);
}
static float bb_perimeter(const float* a) { float wx = a[2] - a[0], wy = a[3] - a[1]; return 2.0f * (wx + wy); }
static int t_alloc(Sim* s) {
  if (s->t_free == T_NULL) {                       /* pool exhausted: new nodes continue the id sequence */
    int old = s->t_cap;
    s->t_cap = old ? 2 * old : 16;
    s->tn = (struct TNode*)realloc(s->tn, sizeof(struct TNode) * s->t_cap);
    s->veh_of = (int*)realloc(s->veh_of, sizeof(int) * s->t_cap);
    for (int i = old; i < s->t_cap; ++i) { s->tn[i].parent = i + 1 < s->t_cap ? i + 1 : T_NULL; s->tn[i].height = -1; s->veh_of[i] = -1; }
    s->t_free = old;
  }
  int id = s->t_free;
  s->t_free = s->tn[id].parent;
  s->tn[id].parent = s->tn[id].c1 = s->tn[id].c2 = T_NULL; s->tn[id].height = 0; s->veh_of[id] = -1;
  return id;
}
static void t_free_node(Sim* s, int id) { s->tn[id].parent = s->t_free; s->tn[id].height = -1; s->t_free = id; }
static int t_is_leaf(const Sim* s, int id) { return s->tn[id].c1 == T_NULL; }
static int t_balance(Sim* s, int iA) {
  struct TNode* N = s->tn;
  if (t_is_leaf(s, iA) || N[iA].height < 2) return iA;
  int iB = N[iA].c1, iC = N[iA].c2;
  int balance = N[iC].height - N[iB].height;
  if (balance > 1) {                               /* rotate C up */
    int iF = N[iC].c1, iG = N[iC].c2;
    N[iC].c1 = iA; N[iC].parent = N[iA].parent; N[iA].parent = iC;
    if (N[iC].parent != T_NULL) { if (N[N[iC].parent].c1 == iA) N[N[iC].parent].c1 = iC; else N[N[iC].parent].c2 = iC; }
    else s->t_root = iC;
    if (N[iF].height > N[iG].height) {
      N[iC].c2 = iF; N[iA].c2 = iG; N[iG].parent = iA;
      bb_combine(N[iA].bb, N[iB].bb, N[iG].bb); bb_combine(N[iC].bb, N[iA].bb, N[iF].bb);
      N[iA].height = 1 + (N[iB].height > N[iG].height ? N[iB].height : N[iG].height);
      N[iC].height = 1 + (N[iA].height > N[iF].height ? N[iA].height : N[iF].height);
    } else {
      N[iC].c2 = iG; N[iA].c2 = iF; N[iF].parent = iA;
      bb_combine(N[iA].bb, N[iB].bb, N[iF].bb); bb_combine(N[iC].bb, N[iA].bb, N[iG].bb);
      N[iA].height = 1 + (N[iB].height > N[iF].height ? N[iB].height : N[iF].height);
      N[iC].height = 1 + (N[iA].height > N[iG].height ? N[iA].height : N[iG].height);
    }
    return iC;
  }
  if (balance < -1) {                              /* rotate B up */
    int iD = N[iB].c1, iE = N[iB].c2;
    N[iB].c1 = iA; N[iB].parent = N[iA].parent; N[iA].parent = iB;
    if (N[iB].parent != T_NULL) { if (N[N[iB].parent].c1 == iA) N[N[iB].parent].c1 = iB; else N[N[iB].parent].c2 = iB; }
    else s->t_root = iB;
    if (N[iD].height > N[iE].height) {
      N[iB].c2 = iD; N[iA].c1 = iE; N[iE].parent = iA;
      bb_combine(N[iA].bb, N[iC].bb, N[iE].bb); bb_combine(N[iB].bb, N[iA].bb, N[iD].bb);
      N[iA].height = 1 + (N[iC].height > N[iE].height ? N[iC].height : N[iE].height);
      N[iB].height = 1 + (N[iA].height > N[iD].height ? N[iA].height : N[iD].height);
    } else {
      N[iB].c2 = iE; N[iA].c1 = iD; N[iD].parent = iA;
      bb_combine(N[iA].bb, N[iC].bb, N[iD].bb); bb_combine(N[iB].bb, N[iA].bb, N[iE].bb);
      N[iA].height = 1 + (N[iC].height > N[iD].height ? N[iC].height : N[iD].height);
      N[iB].height = 1 + (N[iA].height > N[iE].height ? N[iA].height : N[iE].height);
    }
    return iB;
  }
  return iA;
}
static void t_insert_leaf(Sim* s, int leaf) {
  if (s->t_root == T_NULL) { s->t_root = leaf; s->tn[leaf].parent = T_NULL; return; }
  float lb[4]; memcpy(lb, s->tn[leaf].bb, sizeof(lb));
  int index = s->t_root;
  while (!t_is_leaf(s, index)) {                   /* surface-area heuristic descent */
    int c1 = s->tn[index].c1, c2 = s->tn[index].c2;
    float area = bb_perimeter(s->tn[index].bb);
    float comb[4]; bb_combine(comb, s->tn[index].bb, lb);
    float combined_area = bb_perimeter(comb);
    float cost = 2.0f * combined_area;
    float inheritance = 2.0f * (combined_area - area);
    float cost1, cost2, t[4];
    bb_combine(t, lb, s->tn[c1].bb);
    if (t_is_leaf(s, c1)) cost1 = bb_perimeter(t) + inheritance;
    else { float old_area = bb_perimeter(s->tn[c1].bb), new_area = bb_perimeter(t); cost1 = (new_area - old_area) + inheritance; }
    bb_combine(t, lb, s->tn[c2].bb);
    if (t_is_leaf(s, c2)) cost2 = bb_perimeter(t) + inheritance;
    else { float old_area = bb_perimeter(s->tn[c2].bb), new_area = bb_perimeter(t); cost2 = new_area - old_area + inheritance; }
    if (cost < cost1 && cost < cost2) break;
    index = cost1 < cost2 ? c1 : c2;
  }
  int sibling = index;
  int old_parent = s->tn[sibling].parent;
  int new_parent = t_alloc(s);
  struct TNode* N = s->tn;                          /* (t_alloc may have moved the pool) */
  N[new_parent].parent = old_parent;
  bb_combine(N[new_parent].bb, lb, N[sibling].bb);
  N[new_parent].height = N[sibling].height + 1;
  if (old_parent != T_NULL) { if (N[old_parent].c1 == sibling) N[old_parent].c1 = new_parent; else N[old_parent].c2 = new_parent; }
  else s->t_root = new_parent;
  N[new_parent].c1 = sibling; N[new_parent].c2 = leaf;
  N[sibling].parent = new_parent; N[leaf].parent = new_parent;
  index = N[leaf].parent;
  while (index != T_NULL) {                        /* walk up: rebalance, refit */
    index = t_balance(s, index);
    int c1 = N[index].c1, c2 = N[index].c2;
    N[index].height = 1 + (N[c1].height > N[c2].height ? N[c1].height : N[c2].height);
    bb_combine(N[index].bb, N[c1].bb, N[c2].bb);
    index = N[index].parent;
  }
}
static void t_remove_leaf(Sim* s, int leaf) {
  struct TNode* N = s->tn;
  if (leaf == s->t_root) { s->t_root = T_NULL; return; }
  int parent = N[leaf].parent, grand = N[parent].parent;
  int sibling = N[parent].c1 == leaf ? N[parent].c2 : N[parent].c1;
  if (grand != T_NULL) {
    if (N[grand].c1 == parent) N[grand].c1 = sibling; else N[grand].c2 = sibling;
    N[sibling].parent = grand;
    t_free_node(s, parent);
    int index = grand;
    while (index != T_NULL) {
      index = t_balance(s, index);
      int c1 = N[index].c1, c2 = N[index].c2;
      bb_combine(N[index].bb, N[c1].bb, N[c2].bb);
      N[index].height = 1 + (N[c1].height > N[c2].height ? N[c1].height : N[c2].height);
      index = N[index].parent;
    }
  } else {
    s->t_root = sibling; N[sibling].parent = T_NULL;
    t_free_node(s, parent);
  }
}
static void t_create_proxy(Sim* s, int veh) {      /* fat[veh] already holds the fattened box */
  int id = t_alloc(s);
  memcpy(s->tn[id].bb, s->fat + 4 * veh, 4 * sizeof(float));
  s->veh_of[id] = veh; s->leaf_of[veh] = id;
  t_insert_leaf(s, id);
}

static void move_proxy(Sim* s, int i, const float* aabb, float dx, float dy) {
  float fatn[4] = {aabb[0] - B2_AABB_EXT, aabb[1] - B2_AABB_EXT, aabb[2] + B2_AABB_EXT, aabb[3] + B2_AABB_EXT};
  float ddx = B2_AABB_MULT * dx, ddy = B2_AABB_MULT * dy;
  if (ddx < 0.0f) fatn[0] += ddx; else fatn[2] += ddx;
  if (ddy < 0.0f) fatn[1] += ddy; else fatn[3] += ddy;
  float* tree = s->fat + 4 * i;
  if (aabb_contains(tree, aabb)) {
    float huge[4] = {fatn[0] - 4.0f * B2_AABB_EXT, fatn[1] - 4.0f * B2_AABB_EXT, fatn[2] + 4.0f * B2_AABB_EXT, fatn[3] + 4.0f * B2_AABB_EXT};
    if (aabb_contains(huge, tree)) return;
  }
  t_remove_leaf(s, s->leaf_of[i]);
  memcpy(tree, fatn, sizeof(fatn));
  memcpy(s->tn[s->leaf_of[i]].bb, fatn, sizeof(fatn));
  t_insert_leaf(s, s->leaf_of[i]);
  s->moved[i] = 1;
  buffer_move(s, i);
}
static void synchronize_fixture(Sim* s, int i, Xf xf1, Xf xf2) {
  float a1[4], a2[4], c[4];
  shape_aabb(&s->v[i], xf1, a1); shape_aabb(&s->v[i], xf2, a2);
  c[0] = b2minf(a1[0], a2[0]); c[1] = b2minf(a1[1], a2[1]); c[2] = b2maxf(a1[2], a2[2]); c[3] = b2maxf(a1[3], a2[3]);
  float dx = 0.5f * (a2[0] + a2[2]) - 0.5f * (a1[0] + a1[2]);
  float dy = 0.5f * (a2[1] + a2[3]) - 0.5f * (a1[1] + a1[3]);
  move_proxy(s, i, c, dx, dy);
}
/* b2BroadPhase::UpdatePairs + QueryCallback + b2ContactManager::AddPair: one tree query per buffered proxy; a query
 * reports its hits in the tree's stack order (child2 before child1).  Leaf ids grow with vehicle index (each vehicle takes
 * a fresh leaf, parents recycle among themselves), so "proxyId > queryProxyId" is "o > q". */
static void find_new_contacts(Sim* s) {
  int n = s->n;
  int stack[256];
  for (int k = 0; k < s->n_move; ++k) {
    int q = s->move_buf[k];
    const float* fq = s->fat + 4 * q;
    int sc = 0;
    stack[sc++] = s->t_root;
    while (sc > 0) {
      int id = stack[--sc];
      if (id == T_NULL) continue;
      if (!aabb_overlap(s->tn[id].bb, fq)) continue;
      if (!t_is_leaf(s, id)) { stack[sc++] = s->tn[id].c1; stack[sc++] = s->tn[id].c2; continue; }
      int o = s->veh_of[id];
      if (o == q) continue;
      if (s->moved[o] && o > q) continue;
      int i = o < q ? o : q, j = o < q ? q : o;
      if (s->c_exists[i * n + j]) continue;
      s->c_exists[i * n + j] = 1;
      s->c_stamp[i * n + j] = ++s->stamp;
      Manifold* m = &s->man[i * n + j];
      memset(m, 0, sizeof(*m));
    }
  }
  for (int k = 0; k < s->n_move; ++k) s->moved[s->move_buf[k]] = 0;
  s->n_move = 0;
}

/* b2ContactManager::Collide + b2Contact::Update over the contact list (newest contact first) */
static int cmp_stamp_desc(const void* a, const void* b) { return ((const int*)b)[0] - ((const int*)a)[0]; }
static void contacts_update(Sim* s) {
  int n = s->n, nc = 0;
  int* order = (int*)malloc(sizeof(int) * 2 * ((size_t)n * n / 2 + 1));
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (s->c_exists[i * n + j]) { order[2 * nc] = s->c_stamp[i * n + j]; order[2 * nc + 1] = i * n + j; ++nc; }
  qsort(order, nc, 2 * sizeof(int), cmp_stamp_desc);
  for (int c = 0; c < nc; ++c) {
    int ij = order[2 * c + 1], i = ij / n, j = ij % n;
    Veh *A = &s->v[i], *B = &s->v[j];
    if (!A->awake && !B->awake) continue;
    Manifold* m = &s->man[ij];
    if (!aabb_overlap(s->fat + 4 * i, s->fat + 4 * j)) {    /* the fat AABBs ceased to overlap: contact destroyed */
      s->c_exists[ij] = 0;
      memset(m, 0, sizeof(*m));
      continue;
    }
    Manifold old = *m;
    Box bA = box_of(A), bB = box_of(B);
    collide_boxes(m, &bA, body_xf(A), &bB, body_xf(B));
    int touching = m->count > 0;
    for (int k = 0; k < m->count; ++k) {
      MPoint* mp2 = &m->p[k];
      mp2->ni = 0.0f; mp2->ti = 0.0f;
      for (int l = 0; l < old.count; ++l)
        if (old.p[l].id == mp2->id) { mp2->ni = old.p[l].ni; mp2->ti = old.p[l].ti; break; }
    }
    if (touching != old.touching) { set_awake_true(A); set_awake_true(B); }
    m->touching = touching;
  }
  free(order);
}

typedef struct {                      /* b2ContactVelocityConstraint + b2ContactPositionConstraint of one contact */
  int ia, ib, count, vcount;          /* island body indices; manifold points; points the velocity solver uses */
  Manifold* m;
  V2 normal, rA[2], rB[2];
  float nmass[2], tmass[2], nimp[2], timp[2];
  float K[4], NM[4];                  /* ex.x, ex.y, ey.x, ey.y */
} Constraint;

static void island_solve(Sim* s, const int* bodies, int nb, Manifold** contacts, const int* cA, const int* cB, int nc, float h,
                         float dt_ratio) {
  V2* pc = (V2*)malloc(sizeof(V2) * nb); float* pa = (float*)malloc(sizeof(float) * nb);
  V2* vv = (V2*)malloc(sizeof(V2) * nb); float* vw = (float*)malloc(sizeof(float) * nb);
  Constraint* C = (Constraint*)malloc(sizeof(Constraint) * (nc > 0 ? nc : 1));
  for (int i = 0; i < nb; ++i) {      /* gravity, forces, torque, damping are zero: v += +0 (turns a -0 into +0), then * 1.0f */
    const Veh* b = &s->v[bodies[i]];
    s->sweep0[3 * bodies[i]] = b->cx; s->sweep0[3 * bodies[i] + 1] = b->cy; s->sweep0[3 * bodies[i] + 2] = b->a;
    pc[i] = v2(b->cx, b->cy); pa[i] = b->a; vv[i] = v2(b->vx + 0.0f, b->vy + 0.0f); vw[i] = b->w + 0.0f;
  }
  const float friction = sqrtf(0.2f * 0.2f);             /* b2MixFriction of two default fixtures */
  /* ---- constructor + InitializeVelocityConstraints */
  for (int c = 0; c < nc; ++c) {
    Constraint* k = &C[c];
    k->m = contacts[c]; k->ia = cA[c]; k->ib = cB[c]; k->count = k->vcount = k->m->count;
    const Veh *bA = &s->v[bodies[k->ia]], *bB = &s->v[bodies[k->ib]];
    float mA = bA->inv_mass, mB = bB->inv_mass, iA = bA->inv_i, iB = bB->inv_i;
    for (int j = 0; j < k->count; ++j) { k->nimp[j] = dt_ratio * k->m->p[j].ni; k->timp[j] = dt_ratio * k->m->p[j].ti; }
    V2 cAv = pc[k->ia], cBv = pc[k->ib];
    Xf xfA, xfB;
    xfA.q.s = sinf(pa[k->ia]); xfA.q.c = cosf(pa[k->ia]);
    xfB.q.s = sinf(pa[k->ib]); xfB.q.c = cosf(pa[k->ib]);
    { V2 r = rot_mul(xfA.q, v2(bA->lcx, bA->lcy)); xfA.p = v2(cAv.x - r.x, cAv.y - r.y); }
    { V2 r = rot_mul(xfB.q, v2(bB->lcx, bB->lcy)); xfB.p = v2(cBv.x - r.x, cBv.y - r.y); }
    /* b2WorldManifold::Initialize */
    V2 wn, wp[2];
    if (k->m->type == M_FACE_A) {
      wn = rot_mul(xfA.q, v2(k->m->lnx, k->m->lny));
      V2 pp = xf_mul(xfA, v2(k->m->lpx, k->m->lpy));
      for (int j = 0; j < k->count; ++j) {
        V2 cp = xf_mul(xfB, v2(k->m->p[j].lx, k->m->p[j].ly));
        float t = B2_POLY_RADIUS - dot2(v2(cp.x - pp.x, cp.y - pp.y), wn);
        V2 a = v2(cp.x + t * wn.x, cp.y + t * wn.y);
        V2 b = v2(cp.x - B2_POLY_RADIUS * wn.x, cp.y - B2_POLY_RADIUS * wn.y);
        wp[j] = v2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
      }
    } else {
      wn = rot_mul(xfB.q, v2(k->m->lnx, k->m->lny));
      V2 pp = xf_mul(xfB, v2(k->m->lpx, k->m->lpy));
      for (int j = 0; j < k->count; ++j) {
        V2 cp = xf_mul(xfA, v2(k->m->p[j].lx, k->m->p[j].ly));
        float t = B2_POLY_RADIUS - dot2(v2(cp.x - pp.x, cp.y - pp.y), wn);
        V2 b = v2(cp.x + t * wn.x, cp.y + t * wn.y);
        V2 a = v2(cp.x - B2_POLY_RADIUS * wn.x, cp.y - B2_POLY_RADIUS * wn.y);
        wp[j] = v2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
      }
      wn = v2(-wn.x, -wn.y);
    }
    k->normal = wn;
    for (int j = 0; j < k->count; ++j) {
      k->rA[j] = v2(wp[j].x - cAv.x, wp[j].y - cAv.y);
      k->rB[j] = v2(wp[j].x - cBv.x, wp[j].y - cBv.y);
      float rnA = crossvv(k->rA[j], wn), rnB = crossvv(k->rB[j], wn);
      float kn = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
      k->nmass[j] = kn > 0.0f ? 1.0f / kn : 0.0f;
      V2 tg = cross_vs(wn, 1.0f);
      float rtA = crossvv(k->rA[j], tg), rtB = crossvv(k->rB[j], tg);
      float kt = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
      k->tmass[j] = kt > 0.0f ? 1.0f / kt : 0.0f;
      /* restitution 0: velocityBias = -0 * vRel = 0 in every branch */
    }
    if (k->vcount == 2) {
      float rn1A = crossvv(k->rA[0], wn), rn1B = crossvv(k->rB[0], wn);
      float rn2A = crossvv(k->rA[1], wn), rn2B = crossvv(k->rB[1], wn);
      float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
      float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
      float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
      if (k11 * k11 < 1000.0f * (k11 * k22 - k12 * k12)) {
        k->K[0] = k11; k->K[1] = k12; k->K[2] = k12; k->K[3] = k22;
        float a = k11, b = k12, c2 = k12, d = k22;                /* b2Mat22::GetInverse */
        float det = a * d - b * c2;
        if (det != 0.0f) det = 1.0f / det;
        k->NM[0] = det * d; k->NM[2] = -det * b; k->NM[1] = -det * c2; k->NM[3] = det * a;
      } else {
        k->vcount = 1;
      }
    }
  }
  /* ---- WarmStart */
  for (int c = 0; c < nc; ++c) {
    Constraint* k = &C[c];
    const Veh *bA = &s->v[bodies[k->ia]], *bB = &s->v[bodies[k->ib]];
    float mA = bA->inv_mass, mB = bB->inv_mass, iA = bA->inv_i, iB = bB->inv_i;
    V2 vA = vv[k->ia], vB = vv[k->ib]; float wA = vw[k->ia], wB = vw[k->ib];
    V2 nrm = k->normal, tg = cross_vs(nrm, 1.0f);
    for (int j = 0; j < k->vcount; ++j) {
      V2 P = v2(k->nimp[j] * nrm.x + k->timp[j] * tg.x, k->nimp[j] * nrm.y + k->timp[j] * tg.y);
      wA -= iA * crossvv(k->rA[j], P);
      vA.x -= mA * P.x; vA.y -= mA * P.y;
      wB += iB * crossvv(k->rB[j], P);
      vB.x += mB * P.x; vB.y += mB * P.y;
    }
    vv[k->ia] = vA; vw[k->ia] = wA; vv[k->ib] = vB; vw[k->ib] = wB;
  }
  /* ---- 8 velocity iterations */
  for (int it = 0; it < 8; ++it)
    for (int c = 0; c < nc; ++c) {
      Constraint* k = &C[c];
      const Veh *bA = &s->v[bodies[k->ia]], *bB = &s->v[bodies[k->ib]];
      float mA = bA->inv_mass, mB = bB->inv_mass, iA = bA->inv_i, iB = bB->inv_i;
      V2 vA = vv[k->ia], vB = vv[k->ib]; float wA = vw[k->ia], wB = vw[k->ib];
      V2 nrm = k->normal, tg = cross_vs(nrm, 1.0f);
#define REL_V(j) { V2 cb = cross_sv(wB, k->rB[j]), ca = cross_sv(wA, k->rA[j]); dv = v2(vB.x + cb.x - vA.x - ca.x, vB.y + cb.y - vA.y - ca.y); }
#define APPLY(P, j) { vA.x -= mA * P.x; vA.y -= mA * P.y; wA -= iA * crossvv(k->rA[j], P); vB.x += mB * P.x; vB.y += mB * P.y; wB += iB * crossvv(k->rB[j], P); }
      for (int j = 0; j < k->vcount; ++j) {                /* friction first */
        V2 dv; REL_V(j)
        float vt = dot2(dv, tg) - 0.0f;
        float lambda = k->tmass[j] * (-vt);
        float max_f = friction * k->nimp[j];
        float ni = b2maxf(-max_f, b2minf(k->timp[j] + lambda, max_f));     /* b2Clamp = b2Max(low, b2Min(a, high)) */
        lambda = ni - k->timp[j];
        k->timp[j] = ni;
        V2 P = v2(lambda * tg.x, lambda * tg.y);
        APPLY(P, j)
      }
      if (k->vcount == 1) {
        for (int j = 0; j < 1; ++j) {
          V2 dv; REL_V(j)
          float vn = dot2(dv, nrm);
          float lambda = -k->nmass[j] * (vn - 0.0f);
          float ni = b2maxf(k->nimp[j] + lambda, 0.0f);
          lambda = ni - k->nimp[j];
          k->nimp[j] = ni;
          V2 P = v2(lambda * nrm.x, lambda * nrm.y);
          APPLY(P, j)
        }
      } else {                                             /* block solver (2-point LCP) */
        V2 a = v2(k->nimp[0], k->nimp[1]);
        V2 dv; REL_V(0) V2 dv1 = dv; REL_V(1) V2 dv2 = dv;
        float vn1 = dot2(dv1, nrm), vn2 = dot2(dv2, nrm);
        V2 b = v2(vn1 - 0.0f, vn2 - 0.0f);
        { V2 Ka = v2(k->K[0] * a.x + k->K[2] * a.y, k->K[1] * a.x + k->K[3] * a.y); b.x -= Ka.x; b.y -= Ka.y; }
        V2 x; int done = 0;
#define BLOCK_APPLY { V2 d = v2(x.x - a.x, x.y - a.y); V2 P1 = v2(d.x * nrm.x, d.x * nrm.y), P2 = v2(d.y * nrm.x, d.y * nrm.y); \
          vA.x -= mA * (P1.x + P2.x); vA.y -= mA * (P1.y + P2.y); wA -= iA * (crossvv(k->rA[0], P1) + crossvv(k->rA[1], P2)); \
          vB.x += mB * (P1.x + P2.x); vB.y += mB * (P1.y + P2.y); wB += iB * (crossvv(k->rB[0], P1) + crossvv(k->rB[1], P2)); \
          k->nimp[0] = x.x; k->nimp[1] = x.y; done = 1; }
        { V2 t = v2(k->NM[0] * b.x + k->NM[2] * b.y, k->NM[1] * b.x + k->NM[3] * b.y); x = v2(-t.x, -t.y); }
        if (x.x >= 0.0f && x.y >= 0.0f) BLOCK_APPLY
        if (!done) {
          x.x = -k->nmass[0] * b.x; x.y = 0.0f;
          vn2 = k->K[1] * x.x + b.y;
          if (x.x >= 0.0f && vn2 >= 0.0f) BLOCK_APPLY
        }
        if (!done) {
          x.x = 0.0f; x.y = -k->nmass[1] * b.y;
          vn1 = k->K[2] * x.y + b.x;
          if (x.y >= 0.0f && vn1 >= 0.0f) BLOCK_APPLY
        }
        if (!done) {
          x.x = 0.0f; x.y = 0.0f;
          vn1 = b.x; vn2 = b.y;
          if (vn1 >= 0.0f && vn2 >= 0.0f) BLOCK_APPLY
        }
      }
      vv[k->ia] = vA; vw[k->ia] = wA; vv[k->ib] = vB; vw[k->ib] = wB;
    }
  /* ---- StoreImpulses */
  for (int c = 0; c < nc; ++c)
    for (int j = 0; j < C[c].vcount; ++j) { C[c].m->p[j].ni = C[c].nimp[j]; C[c].m->p[j].ti = C[c].timp[j]; }
  /* ---- integrate positions */
  for (int i = 0; i < nb; ++i) {
    V2 v = vv[i]; float w = vw[i];
    float tx = h * v.x, ty = h * v.y;
    if (tx * tx + ty * ty > B2_MAXTRANSLATION * B2_MAXTRANSLATION) {
      float ratio = B2_MAXTRANSLATION / sqrtf(tx * tx + ty * ty);
      v.x *= ratio; v.y *= ratio;
    }
    float rot = h * w;
    if (rot * rot > B2_MAXROTATION * B2_MAXROTATION) {
      float ratio = B2_MAXROTATION / fabsf(rot);
      w *= ratio;
    }
    pc[i].x += h * v.x; pc[i].y += h * v.y;
    pa[i] += h * w;
    vv[i] = v; vw[i] = w;
  }
  /* ---- 3 position iterations */
  int position_solved = 0;
  for (int it = 0; it < 3; ++it) {
    float min_sep = 0.0f;
    for (int c = 0; c < nc; ++c) {
      Constraint* k = &C[c];
      const Veh *bA = &s->v[bodies[k->ia]], *bB = &s->v[bodies[k->ib]];
      float mA = bA->inv_mass, mB = bB->inv_mass, iA = bA->inv_i, iB = bB->inv_i;
      V2 cAv = pc[k->ia], cBv = pc[k->ib]; float aA = pa[k->ia], aB = pa[k->ib];
      for (int j = 0; j < k->count; ++j) {
        Xf xfA, xfB;
        xfA.q.s = sinf(aA); xfA.q.c = cosf(aA); xfB.q.s = sinf(aB); xfB.q.c = cosf(aB);
        { V2 r = rot_mul(xfA.q, v2(bA->lcx, bA->lcy)); xfA.p = v2(cAv.x - r.x, cAv.y - r.y); }
        { V2 r = rot_mul(xfB.q, v2(bB->lcx, bB->lcy)); xfB.p = v2(cBv.x - r.x, cBv.y - r.y); }
        V2 nrm, point; float separation;
        if (k->m->type == M_FACE_A) {
          nrm = rot_mul(xfA.q, v2(k->m->lnx, k->m->lny));
          V2 pp = xf_mul(xfA, v2(k->m->lpx, k->m->lpy));
          V2 cp = xf_mul(xfB, v2(k->m->p[j].lx, k->m->p[j].ly));
          separation = dot2(v2(cp.x - pp.x, cp.y - pp.y), nrm) - B2_POLY_RADIUS - B2_POLY_RADIUS;
          point = cp;
        } else {
          nrm = rot_mul(xfB.q, v2(k->m->lnx, k->m->lny));
          V2 pp = xf_mul(xfB, v2(k->m->lpx, k->m->lpy));
          V2 cp = xf_mul(xfA, v2(k->m->p[j].lx, k->m->p[j].ly));
          separation = dot2(v2(cp.x - pp.x, cp.y - pp.y), nrm) - B2_POLY_RADIUS - B2_POLY_RADIUS;
          point = cp;
          nrm = v2(-nrm.x, -nrm.y);
        }
        V2 rA = v2(point.x - cAv.x, point.y - cAv.y), rB = v2(point.x - cBv.x, point.y - cBv.y);
        min_sep = b2minf(min_sep, separation);
        float Cc = b2maxf(-B2_MAX_LIN_CORR, b2minf(B2_BAUMGARTE * (separation + B2_LINEAR_SLOP), 0.0f));
        float rnA = crossvv(rA, nrm), rnB = crossvv(rB, nrm);
        float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
        float impulse = K > 0.0f ? -Cc / K : 0.0f;
        V2 P = v2(impulse * nrm.x, impulse * nrm.y);
        cAv.x -= mA * P.x; cAv.y -= mA * P.y; aA -= iA * crossvv(rA, P);
        cBv.x += mB * P.x; cBv.y += mB * P.y; aB += iB * crossvv(rB, P);
      }
      pc[k->ia] = cAv; pa[k->ia] = aA; pc[k->ib] = cBv; pa[k->ib] = aB;
    }
    if (min_sep >= -3.0f * B2_LINEAR_SLOP) { position_solved = 1; break; }
  }
  /* ---- copy back, SynchronizeTransform, sleep */
  float min_sleep = B2_FLT_MAX;
  for (int i = 0; i < nb; ++i) {
    Veh* b = &s->v[bodies[i]];
    b->cx = pc[i].x; b->cy = pc[i].y; b->a = pa[i]; b->vx = vv[i].x; b->vy = vv[i].y; b->w = vw[i];
    float qs = sinf(b->a), qc = cosf(b->a);
    b->px = b->cx - (qc * b->lcx - qs * b->lcy);
    b->py = b->cy - (qs * b->lcx + qc * b->lcy);
    if (b->w * b->w > B2_ANGSLEEPTOL * B2_ANGSLEEPTOL || b->vx * b->vx + b->vy * b->vy > B2_LINSLEEPTOL * B2_LINSLEEPTOL) {
      b->sleep_time = 0.0f; min_sleep = 0.0f;
    } else {
      b->sleep_time += h; min_sleep = b2minf(min_sleep, b->sleep_time);
    }
  }
  if (min_sleep >= B2_TIMETOSLEEP && position_solved)
    for (int i = 0; i < nb; ++i) {
      Veh* b = &s->v[bodies[i]];
      b->awake = 0; b->sleep_time = 0.0f; b->vx = b->vy = 0.0f; b->w = 0.0f;
    }
  free(pc); free(pa); free(vv); free(vw); free(C);
}

/* b2World::Step for this world: Collide, Solve (islands by DFS over touching contacts), m_inv_dt0 */
static void world_step(Sim* s, float dt) {
  int n = s->n;
  if (s->new_contacts) { find_new_contacts(s); s->new_contacts = 0; }
  contacts_update(s);
  float inv_dt = dt > 0.0f ? 1.0f / dt : 0.0f;
  float dt_ratio = s->inv_dt0 * dt;
  char* in_island = (char*)calloc(n > 0 ? n : 1, 1);
  char* was_in_island = (char*)calloc(n > 0 ? n : 1, 1);
  char* c_flag = (char*)calloc((size_t)n * n + 1, 1);
  int* stack = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
  int* bodies = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
  int* isl_index = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
  int* edges = (int*)malloc(sizeof(int) * 2 * (n > 0 ? n : 1));
  Manifold** contacts = (Manifold**)malloc(sizeof(Manifold*) * ((size_t)n * n / 2 + 1));
  int* cA = (int*)malloc(sizeof(int) * ((size_t)n * n / 2 + 1));
  int* cB = (int*)malloc(sizeof(int) * ((size_t)n * n / 2 + 1));
  for (int seed = n - 1; seed >= 0; --seed) {            /* m_bodyList: newest body first */
    if (in_island[seed] || !s->v[seed].awake) continue;
    int nb = 0, nc = 0, sc = 0;
    stack[sc++] = seed; in_island[seed] = 1;
    while (sc > 0) {
      int b = stack[--sc];
      isl_index[b] = nb; bodies[nb++] = b;
      s->v[b].awake = 1;                                  /* wake without resetting the sleep timer */
      int ne = 0;                                         /* b's contact edges, newest contact first */
      for (int o = 0; o < n; ++o) {
        if (o == b) continue;
        int i = b < o ? b : o, j = b < o ? o : b;
        if (s->c_exists[i * n + j]) { edges[2 * ne] = s->c_stamp[i * n + j]; edges[2 * ne + 1] = o; ++ne; }
      }
      qsort(edges, ne, 2 * sizeof(int), cmp_stamp_desc);
      for (int e = 0; e < ne; ++e) {
        int o = edges[2 * e + 1];
        int i = b < o ? b : o, j = b < o ? o : b;
        Manifold* m = &s->man[i * n + j];
        if (c_flag[i * n + j] || !m->touching) continue;
        c_flag[i * n + j] = 1;
        contacts[nc] = m; cA[nc] = i; cB[nc] = j; ++nc;
        if (in_island[o]) continue;
        stack[sc++] = o; in_island[o] = 1;
      }
    }
    for (int c = 0; c < nc; ++c) { cA[c] = isl_index[cA[c]]; cB[c] = isl_index[cB[c]]; }
    island_solve(s, bodies, nb, contacts, cA, cB, nc, dt, dt_ratio);
  }
  /* SynchronizeFixtures of every body that was in an island (m_bodyList order), then FindNewContacts */
  for (int b = n - 1; b >= 0; --b) {
    if (!in_island[b]) continue;
    Veh* v = &s->v[b];
    Xf xf2 = body_xf(v);
    if (v->awake) {
      Xf xf1;
      xf1.q.s = sinf(s->sweep0[3 * b + 2]); xf1.q.c = cosf(s->sweep0[3 * b + 2]);
      V2 r = rot_mul(xf1.q, v2(v->lcx, v->lcy));
      xf1.p = v2(s->sweep0[3 * b] - r.x, s->sweep0[3 * b + 1] - r.y);
      synchronize_fixture(s, b, xf1, xf2);
    } else {
      synchronize_fixture(s, b, xf2, xf2);
    }
  }
  find_new_contacts(s);
  s->inv_dt0 = inv_dt;
  free(in_island); free(was_in_island); free(c_flag); free(stack); free(bodies); free(isl_index); free(edges);
  free(contacts); free(cA); free(cB);
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
  s->man = (Manifold*)calloc((size_t)n * n + 1, sizeof(Manifold));
  s->inv_dt0 = 0.0f;
  s->fat = (float*)calloc((size_t)4 * n + 4, sizeof(float));
  s->moved = (char*)calloc(n + 1, 1);
  s->c_exists = (char*)calloc((size_t)n * n + 1, 1);
  s->c_stamp = (int*)calloc((size_t)n * n + 1, sizeof(int));
  s->sweep0 = (float*)calloc((size_t)3 * n + 3, sizeof(float));
  s->tn = NULL; s->veh_of = NULL; s->t_cap = 0; s->t_root = T_NULL; s->t_free = T_NULL;
  s->leaf_of = (int*)calloc(n + 1, sizeof(int));
  s->move_buf = NULL; s->n_move = s->cap_move = 0; s->stamp = 0; s->new_contacts = 1;
  s->segs = (float*)malloc(sizeof(float) * 4 * (n_seg > 0 ? n_seg : 1));
  if (n_seg > 0) memcpy(s->segs, segs, sizeof(float) * 4 * n_seg);
  for (int i = 0; i < n; ++i) {
    Veh* v = &s->v[i];
    v->length = length[i]; v->width = width[i];
    v->px = x[i]; v->py = y[i]; v->heading = heading[i]; v->speed = speed[i];
    local_center(v->width, v->length, &v->lcx, &v->lcy, &v->inv_mass, &v->inv_i);
    {                                          /* CreateFixture on the body at the origin: b2DynamicTree::CreateProxy */
      v->px = v->py = 0.f; v->a = 0.f;
      float bb[4];
      shape_aabb(v, body_xf(v), bb);
      s->fat[4 * i] = bb[0] - B2_AABB_EXT; s->fat[4 * i + 1] = bb[1] - B2_AABB_EXT;
      s->fat[4 * i + 2] = bb[2] + B2_AABB_EXT; s->fat[4 * i + 3] = bb[3] + B2_AABB_EXT;
      t_create_proxy(s, i);
      s->moved[i] = 1; buffer_move(s, i);
    }
    set_transform(v, 0.f, 0.f, (float)((double)v->heading - M_PI * 0.5f));   /* SetAngle, vehicle.cc:168 */
    synchronize_fixture(s, i, body_xf(v), body_xf(v));
    set_transform(v, x[i], y[i], v->a);                                      /* SetPosition, vehicle.cc:169 */
    synchronize_fixture(s, i, body_xf(v), body_xf(v));
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
  synchronize_fixture((Sim*)h, i, body_xf(v), body_xf(v));
  ((Sim*)h)->new_contacts = 1;
}

void orasim_step(void* h, float dt) {
  Sim* s = (Sim*)h;
  for (int i = 0; i < s->n; ++i) freecar_step(&s->v[i], dt);
  world_step(s, dt);
  for (int i = 0; i < s->n; ++i) {
    Veh* v = &s->v[i];
    v->coll_veh = v->coll_edge = 0;             /* position_ <- m_xf.p (already in px,py) */
    if (!v->ex_on) {
      v->speed = sqrtf(v->vx * v->vx + v->vy * v->vy);
      v->heading = (float)((double)v->a + M_PI * 0.5f);
    } else {
      /* Scenario::Step for an expert-controlled object (nocturne/cpp/src/scenario.cc:276-283): the three Vehicle setters, in order.
       * set_position (vehicle.cc:82-87 -> BaseCar::SetPosition -> b2Body::SetTransform at the current angle): proxy synchronised,
       * new contacts looked for at the top of the next world step; set_heading (vehicle.cc:89-94 -> SetAngle(heading - pi/2) ->
       * SetTransform at the current position); set_speed (vehicle.cc:96-105 -> b2Body::SetLinearVelocity, which wakes the body
       * when the velocity is not zero; Object::ClipSpeed clips at max_speed_ = the float maximum: a no-op). */
      set_transform(v, v->ex_x, v->ex_y, v->a);
      synchronize_fixture(s, i, body_xf(v), body_xf(v));
      v->heading = v->ex_heading;
      set_transform(v, v->px, v->py, (float)((double)v->heading - M_PI * 0.5f));
      synchronize_fixture(s, i, body_xf(v), body_xf(v));
      s->new_contacts = 1;
      v->speed = v->ex_speed;
      {
        float c = cosf(v->heading), sn = sinf(v->heading);
        float nvx = v->ex_speed * c, nvy = v->ex_speed * sn;
        if (nvx * nvx + nvy * nvy > 0.0f) set_awake_true(v);
        v->vx = nvx; v->vy = nvy;
      }
      v->ex_on = 0;
    }
  }
  update_collision(s);
}

void orasim_set_expert(void* h, int i, float x, float y, float heading, float speed) {
  Veh* v = &((Sim*)h)->v[i];
  v->ex_on = 1; v->ex_x = x; v->ex_y = y; v->ex_heading = heading; v->ex_speed = speed;
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
  free(s->v); free(s->segs); free(s->man); free(s->fat); free(s->moved); free(s->c_exists); free(s->c_stamp);
  free(s->sweep0); free(s->move_buf); free(s->tn); free(s->veh_of); free(s->leaf_of); free(s);
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
