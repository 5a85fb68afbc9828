// TEST INFRASTRUCTURE — harness around the REAL reference physics + geometry sources.
//
// Compiled by oracle/Makefile from the sources where they lie under /root/reference
// (nocturne/cpp/src/physics/*.cpp, the vendored+patched Box2D 2.4.1, nocturne/cpp/src/geometry/*.cc)
// into oracle/_ref/libref_sim.so.  No reference source is copied into this repository.
//
// nocturne_core (vehicle.cc / object.cc / scenario.cc) cannot be compiled here (SFML headers are
// absent), so the ~60 lines of glue those files put around the physics and geometry libraries are
// restated below, each block citing what it follows:
//   Vehicle::CreatePhysicsBody        nocturne/cpp/src/vehicle.cc:137-179
//   Vehicle::set_acceleration/brake/set_steering/set_position   vehicle.cc:75-135
//   Scenario::Step / Vehicle::Step    scenario.cc:266-292, vehicle.cc:25-55 (expert_control objects: scenario.cc:276-283,
//                                     Vehicle::set_position / set_heading / set_speed vehicle.cc:75-105)
//   Object::BoundingPolygon           object.cc:14-28
//   Scenario::UpdateCollision         scenario.cc:294-328  (BVH candidates == strict AABB overlap,
//                                     bvh.h:181-193 + aabb.h:47-50, evaluated brute force)
//   Object::Velocity                  object.h:152-154
#include <cmath>
#include <cstdint>
#include <vector>

#include "FreeCar.h"
#include "PhysicsSimulation.h"
#include "Singletons.h"
#include "geometry/aabb.h"
#include "geometry/intersection.h"
#include "geometry/line_segment.h"
#include "geometry/polygon.h"
#include "geometry/vector_2d.h"

using nocturne::geometry::AABB;
using nocturne::geometry::ConvexPolygon;
using nocturne::geometry::LineSegment;
using nocturne::geometry::Vector2D;

namespace {

struct Veh {
  float length, width;
  Vector2D position;
  float heading, speed;
  physics::FreeCar* car;
  bool coll_veh, coll_edge;
  bool expert = false;            // Object::expert_control_ for the next step, with the logged state of that step
  float ex_x = 0.f, ex_y = 0.f, ex_heading = 0.f, ex_speed = 0.f;
};

struct Sim {
  std::vector<Veh> vehs;
  std::vector<LineSegment> segs;
};

ConvexPolygon BoundingPolygon(const Veh& v) {  // object.cc:14-28
  const Vector2D p0 = Vector2D(v.length * 0.5f, v.width * 0.5f).Rotate(v.heading) + v.position;
  const Vector2D p1 = Vector2D(-v.length * 0.5f, v.width * 0.5f).Rotate(v.heading) + v.position;
  const Vector2D p2 = Vector2D(-v.length * 0.5f, -v.width * 0.5f).Rotate(v.heading) + v.position;
  const Vector2D p3 = Vector2D(v.length * 0.5f, -v.width * 0.5f).Rotate(v.heading) + v.position;
  return ConvexPolygon({p0, p1, p2, p3});
}

void UpdateCollision(Sim* s) {  // scenario.cc:294-328
  const int n = static_cast<int>(s->vehs.size());
  std::vector<ConvexPolygon> polys;
  std::vector<AABB> boxes;
  polys.reserve(n);
  for (int i = 0; i < n; ++i) {
    polys.push_back(BoundingPolygon(s->vehs[i]));
    boxes.push_back(polys.back().GetAABB());
  }
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) {
      if (i == j) continue;
      if (!boxes[i].Intersects(boxes[j])) continue;  // BVH candidate predicate
      if (polys[i].Intersects(polys[j])) s->vehs[i].coll_veh = true;
    }
  }
  for (int i = 0; i < n; ++i) {
    for (const LineSegment& seg : s->segs) {
      if (!boxes[i].Intersects(seg.GetAABB())) continue;
      if (nocturne::geometry::Intersects(polys[i], seg)) s->vehs[i].coll_edge = true;
    }
  }
}

// BaseCar::m_Body is protected; a derived-class pointer-to-member is the standard-conforming peek.
struct BodyPeek : physics::BaseCar {
  static b2Body* Get(physics::BaseCar* c) { return c->*(&BodyPeek::m_Body); }
};

}  // namespace

extern "C" {

void* refsim_create(int n, const float* length, const float* width, const float* x, const float* y,
                    const float* heading, const float* speed, int n_seg, const float* segs) {
  physics::GetPhysicsSimulation()->DeleteScene();  // process-global world (Singletons.cpp:5-25)
  Sim* s = new Sim();
  for (int i = 0; i < n; ++i) {
    Veh v;
    v.length = length[i];
    v.width = width[i];
    v.position = Vector2D(x[i], y[i]);
    v.heading = heading[i];
    v.speed = speed[i];
    v.coll_veh = v.coll_edge = false;
    // vehicle.cc:137-179
    v.car = new physics::FreeCar(v.width, v.length);
    v.car->SetAngle(v.heading - M_PI * 0.5f);
    v.car->SetPosition(b2Vec2(v.position.x(), v.position.y()));
    float c = cosf(v.heading);
    float sn = sinf(v.heading);
    b2Vec2 speed_v(v.speed * c, v.speed * sn);
    v.car->SetSpeed(speed_v);
    physics::GetPhysicsSimulation()->AddCar(v.car);
    s->vehs.push_back(v);
  }
  for (int k = 0; k < n_seg; ++k) {
    s->segs.emplace_back(Vector2D(segs[4 * k], segs[4 * k + 1]), Vector2D(segs[4 * k + 2], segs[4 * k + 3]));
  }
  UpdateCollision(s);  // scenario.cc:262-263 (initial flags)
  return s;
}

void refsim_set_action(void* h, int i, double accel, double steer) {
  Sim* s = static_cast<Sim*>(h);
  // autoregressive_policy.py:269-273 compares in Python float (double), pybind narrows to float
  if (accel > 0.0) {
    s->vehs[i].car->Throttle(static_cast<float>(accel));            // vehicle.cc:107-115
  } else {
    s->vehs[i].car->Brake(static_cast<float>(std::fabs(accel)));    // vehicle.cc:128-135
  }
  s->vehs[i].car->Turn(static_cast<float>(steer));                  // vehicle.cc:117-126
}

void refsim_set_position(void* h, int i, float x, float y) {        // vehicle.cc:75-81
  Sim* s = static_cast<Sim*>(h);
  s->vehs[i].position = Vector2D(x, y);
  s->vehs[i].car->SetPosition(b2Vec2(x, y));
}

void refsim_step(void* h, float dt) {                               // scenario.cc:266-292
  Sim* s = static_cast<Sim*>(h);
  physics::GetPhysicsSimulation()->Step(dt);
  for (Veh& v : s->vehs) {
    v.coll_veh = v.coll_edge = false;                               // ResetCollision
    if (!v.expert) {
      b2Vec2 pos = v.car->GetPosition();                            // vehicle.cc:45-55
      v.position = Vector2D(pos.x, pos.y);
      v.speed = v.car->GetSpeed();
      v.heading = v.car->GetAngle() + M_PI * 0.5f;
    } else {
      // scenario.cc:279-283: set_position, set_heading, set_speed of the logged state, each through the Vehicle override
      v.position = Vector2D(v.ex_x, v.ex_y);                        // vehicle.cc:82-87
      v.car->SetPosition(b2Vec2(v.position.x(), v.position.y()));
      v.heading = v.ex_heading;                                     // vehicle.cc:89-94
      v.car->SetAngle(v.heading - M_PI * 0.5f);
      v.speed = v.ex_speed;                                         // vehicle.cc:96-105 (ClipSpeed: max_speed_ is the float maximum)
      float c = cosf(v.heading);
      float sn = sinf(v.heading);
      b2Vec2 speed_v(v.ex_speed * c, v.ex_speed * sn);
      v.car->SetSpeed(speed_v);
      v.expert = false;
    }
  }
  UpdateCollision(s);
}

// the object is expert-controlled in the NEXT step (Object::set_expert_control(true), pybind object.cc:58-59) and the log holds this state for it
void refsim_set_expert(void* h, int i, float x, float y, float heading, float speed) {
  Veh& v = static_cast<Sim*>(h)->vehs[i];
  v.expert = true;
  v.ex_x = x; v.ex_y = y; v.ex_heading = heading; v.ex_speed = speed;
}

// out[n,6] = x, y, heading, speed, vx, vy  (velocity = PolarToVector2D(speed, heading), object.h:152-154)
void refsim_get_state(void* h, float* out, unsigned char* coll_veh, unsigned char* coll_edge) {
  Sim* s = static_cast<Sim*>(h);
  int i = 0;
  for (const Veh& v : s->vehs) {
    const Vector2D vel = nocturne::geometry::PolarToVector2D(v.speed, v.heading);
    out[6 * i + 0] = v.position.x();
    out[6 * i + 1] = v.position.y();
    out[6 * i + 2] = v.heading;
    out[6 * i + 3] = v.speed;
    out[6 * i + 4] = vel.x();
    out[6 * i + 5] = vel.y();
    coll_veh[i] = v.coll_veh;
    coll_edge[i] = v.coll_edge;
    ++i;
  }
}

// raw Box2D body state for debugging parity: out[n,6] = px, py, angle, vx, vy, w
void refsim_get_body(void* h, float* out) {
  Sim* s = static_cast<Sim*>(h);
  int i = 0;
  for (const Veh& v : s->vehs) {
    b2Body* b = BodyPeek::Get(v.car);
    out[6 * i + 0] = b->GetPosition().x;
    out[6 * i + 1] = b->GetPosition().y;
    out[6 * i + 2] = b->GetAngle();
    out[6 * i + 3] = b->GetLinearVelocity().x;
    out[6 * i + 4] = b->GetLinearVelocity().y;
    out[6 * i + 5] = b->GetAngularVelocity();
    ++i;
  }
}

void refsim_destroy(void* h) {
  Sim* s = static_cast<Sim*>(h);
  physics::GetPhysicsSimulation()->DeleteScene();
  delete s;
}

// Geometry KAT entry points (polygon_test.cc:60-86, intersection_test.cc:52-76)
int refgeo_poly_poly(const float* a, int na, const float* b, int nb) {
  std::vector<Vector2D> va, vb;
  for (int i = 0; i < na; ++i) va.emplace_back(a[2 * i], a[2 * i + 1]);
  for (int i = 0; i < nb; ++i) vb.emplace_back(b[2 * i], b[2 * i + 1]);
  return ConvexPolygon(va).Intersects(ConvexPolygon(vb));
}

int refgeo_poly_seg(const float* a, int na, const float* seg) {
  std::vector<Vector2D> va;
  for (int i = 0; i < na; ++i) va.emplace_back(a[2 * i], a[2 * i + 1]);
  return nocturne::geometry::Intersects(ConvexPolygon(va),
                                        LineSegment(Vector2D(seg[0], seg[1]), Vector2D(seg[2], seg[3])));
}

}  // extern "C"
