// Batched per-step vehicle update + collision flags for S independent scenarios (gfx950, float32).
// Compiled with -ffp-contract=off: results must equal the reference's plain IEEE float32 evaluation.
//
// What one launch replaces in the reference (paths relative to /root/reference):
//   policies/autoregressive_policy.py:256-274  act(): token -> (accel, steer), throttle / brake / steer setters,
//                                              teleport of non-existing vehicles to (-1e6,-1e6)
//   nocturne/cpp/src/vehicle.cc:75-135         setters -> FreeCar::Throttle/Brake/Turn
//   nocturne/cpp/src/physics/FreeCar.cpp:66-186 FreeCar controller (DampenSpeed, slip angle beta, steering radius)
//   third_party/box2d/src/dynamics/b2_island.cpp:194-229,279-310,349-392  integrate, translation/rotation clamps
//                                              (patched b2_maxTranslation = 5.0f, b2_common.h:95), auto-sleep;
//                                              contact-free tier: no contact solver (DESIGN.md "scope")
//   nocturne/cpp/src/vehicle.cc:45-55          read-back: position <- xf.p, speed <- |v|, heading <- angle + pi/2
//   nocturne/cpp/src/scenario.cc:266-328       reset flags, vehicle-vehicle SAT and vehicle-road-edge tests with the
//                                              strict-AABB candidate predicate (aabb.h:47-50, bvh.h:181-193)
//   policies/policy.py:68-79                   history append of the 8-float state row
// One workgroup per scenario; lane i < N owns vehicle i (N <= 64: one wavefront of vehicles), all 256 threads share
// the N x N and N x E collision tests.  State is SoA-per-scenario [S, N, ...] so a wave's loads are contiguous.
#include "common.h"

#define PHYS_STRIDE 20
enum { P_CX = 0, P_CY, P_A, P_VX, P_VY, P_W, P_SLEEP, P_AWAKE, P_THR, P_BRK, P_STEER, P_LCX, P_LCY, P_PX, P_PY,
       P_HEADING, P_SPEED };   // P_HEADING/P_SPEED: Object-level heading_/speed_ (what Python reads)

#define B2_PI 3.14159265359f
#define B2_MAXTRANSLATION 5.0f
#define B2_MAXROTATION (0.5f * B2_PI)
#define B2_LINSLEEPTOL 0.01f
#define B2_ANGSLEEPTOL (2.0f / 180.0f * B2_PI)
#define B2_TIMETOSLEEP 0.5f
#define M_PI_D 3.14159265358979323846

struct SimDiscretisation {  // cfgs/dataset/waymo/base.yaml:13-16,41-42
  double min_accel, max_accel, min_steer, max_steer;
  int n_accel, n_steer;
};

__device__ __forceinline__ float dampen(float speed, float target, float damping, float dt) {
  const float red = damping * dt;
  if (speed - target > red) return speed - red;
  if (speed - target < -red) return speed + red;
  return target;
}

// b2PolygonShape::SetAsBox + ComputeMass(density 20) + b2Body::ResetMassData (float32 residue of the centroid)
__device__ void local_center(float width, float length, float* lcx, float* lcy) {
  const float hx = width / 2, hy = length / 2;
  const float vx[4] = {-hx, hx, hx, -hx}, vy[4] = {-hy, -hy, hy, hy};
  float cx = 0.0f, cy = 0.0f, area = 0.0f;
  const float sx = vx[0], sy = vy[0];
  const float k_inv3 = 1.0f / 3.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float e1x = vx[i] - sx, e1y = vy[i] - sy;
    const float e2x = (i + 1 < 4 ? vx[i + 1] : vx[0]) - sx, e2y = (i + 1 < 4 ? vy[i + 1] : vy[0]) - sy;
    const float D = e1x * e2y - e1y * e2x;
    const float ta = 0.5f * D;
    area += ta;
    const float k = ta * k_inv3;
    cx += k * (e1x + e2x);
    cy += k * (e1y + e2y);
  }
  const float mass = 20.f * area;
  const float inv_area = 1.0f / area;
  cx *= inv_area;
  cy *= inv_area;
  const float mcx = cx + sx, mcy = cy + sy;
  const float lx = mass * mcx, ly = mass * mcy;
  const float inv_mass = 1.0f / mass;
  *lcx = lx * inv_mass;
  *lcy = ly * inv_mass;
}

__device__ __forceinline__ void set_transform(float* p, float x, float y, float angle) {
  const float qs = sinf(angle), qc = cosf(angle);
  p[P_PX] = x;
  p[P_PY] = y;
  p[P_A] = angle;
  p[P_CX] = (qc * p[P_LCX] - qs * p[P_LCY]) + x;
  p[P_CY] = (qs * p[P_LCX] + qc * p[P_LCY]) + y;
}

__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

__device__ bool separates(const float* e0, const float* e1, const float* poly) {
  const float dx = e1[0] - e0[0], dy = e1[1] - e0[1];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (cross2(poly[2 * k] - e0[0], poly[2 * k + 1] - e0[1], dx, dy) <= 0.0f) return false;
  return true;
}

__device__ bool box_box(const float* a, const float* b) {   // polygon.cc:84-98
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (separates(a + 2 * k, a + 2 * ((k + 1) & 3), b)) return false;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (separates(b + 2 * k, b + 2 * ((k + 1) & 3), a)) return false;
  return true;
}

__device__ bool box_contains(const float* a, float x, float y) {   // polygon.cc:66-79
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (cross2(x - a[2 * (i - 1)], y - a[2 * (i - 1) + 1], a[2 * i] - a[2 * (i - 1)], a[2 * i + 1] - a[2 * (i - 1) + 1]) > 0.0f)
      return false;
  return cross2(x - a[6], y - a[7], a[0] - a[6], a[1] - a[7]) <= 0.0f;
}

__device__ bool box_seg(const float* a, float s0, float s1, float s2, float s3) {   // intersection.cc:200-232
  if (s0 == s2 && s1 == s3) return box_contains(a, s0, s1);
  const float dx = s2 - s0, dy = s3 - s1;
  float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float cur = cross2(a[2 * k] - s0, a[2 * k + 1] - s1, dx, dy);
    mn = fminf(mn, cur);
    mx = fmaxf(mx, cur);
  }
  if (mx < 0.0f || mn > 0.0f) return false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* e0 = a + 2 * k;
    const float* e1 = a + 2 * ((k + 1) & 3);
    const float cx = e1[0] - e0[0], cy = e1[1] - e0[1];
    const float v0 = cross2(s0 - e0[0], s1 - e0[1], cx, cy);
    const float v1 = cross2(s2 - e0[0], s3 - e0[1], cx, cy);
    if (v0 > 0.0f && v1 > 0.0f) return false;
  }
  return true;
}

// shared by init and step: corners/AABBs into LDS, flags, history row
__device__ void collide_and_record(int s, int N, int E, const float* __restrict__ size, const float* __restrict__ edges,
                                   const unsigned char* __restrict__ exists, float* __restrict__ hist_states,
                                   unsigned char* __restrict__ coll, int t_row, int Tmax1, float (*corner)[8],
                                   float (*box)[4], int* flag_veh, int* flag_edge, const float* px, const float* py,
                                   const float* heading, const float* speed) {
  const int tid = threadIdx.x;
  if (tid < N) {
    const float L = size[((size_t)s * N + tid) * 2 + 0], Wd = size[((size_t)s * N + tid) * 2 + 1];
    const float st = sinf(heading[tid]), ct = cosf(heading[tid]);        // object.cc:14-28
    const float hx[4] = {L * 0.5f, -L * 0.5f, -L * 0.5f, L * 0.5f};
    const float hy[4] = {Wd * 0.5f, Wd * 0.5f, -Wd * 0.5f, -Wd * 0.5f};
    float b0 = 3.402823466e+38f, b1 = 3.402823466e+38f, b2 = -3.402823466e+38f, b3 = -3.402823466e+38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float x = (hx[k] * ct - hy[k] * st) + px[tid];
      const float y = (hx[k] * st + hy[k] * ct) + py[tid];
      corner[tid][2 * k] = x;
      corner[tid][2 * k + 1] = y;
      b0 = fminf(b0, x); b2 = fmaxf(b2, x);
      b1 = fminf(b1, y); b3 = fmaxf(b3, y);
    }
    box[tid][0] = b0; box[tid][1] = b1; box[tid][2] = b2; box[tid][3] = b3;
    flag_veh[tid] = 0;
    flag_edge[tid] = 0;
    // history row (policies/policy.py:68-79): x, y, vx, vy, heading, length, width, existence
    float* row = hist_states + (((size_t)s * N + tid) * Tmax1 + t_row) * 8;
    row[0] = px[tid];
    row[1] = py[tid];
    row[2] = speed[tid] * cosf(heading[tid]);                           // object.h:152-154
    row[3] = speed[tid] * sinf(heading[tid]);
    row[4] = heading[tid];
    row[5] = L;
    row[6] = Wd;
    row[7] = exists[(size_t)s * N + tid] ? 1.0f : 0.0f;
  }
  __syncthreads();
  for (int p = tid; p < N * N; p += blockDim.x) {
    const int i = p / N, j = p - i * N;
    if (i == j) continue;
    if (!(box[i][0] < box[j][2] && box[i][2] > box[j][0] && box[i][1] < box[j][3] && box[i][3] > box[j][1])) continue;
    if (box_box(corner[i], corner[j])) atomicOr(&flag_veh[i], 1);
  }
  const float* eg = edges + (size_t)s * E * 4;
  for (int e = tid; e < E; e += blockDim.x) {
    const f32x4 sg = *reinterpret_cast<const f32x4*>(eg + (size_t)e * 4);
    const float s0 = fminf(sg[0], sg[2]), s1 = fminf(sg[1], sg[3]), s2 = fmaxf(sg[0], sg[2]), s3 = fmaxf(sg[1], sg[3]);
    for (int i = 0; i < N; ++i) {
      if (!(box[i][0] < s2 && box[i][2] > s0 && box[i][1] < s3 && box[i][3] > s1)) continue;
      if (box_seg(corner[i], sg[0], sg[1], sg[2], sg[3])) atomicOr(&flag_edge[i], 1);
    }
  }
  __syncthreads();
  if (tid < N) {
    unsigned char* c = coll + (((size_t)s * N + tid) * Tmax1 + t_row) * 2;
    c[0] = (unsigned char)flag_veh[tid];
    c[1] = (unsigned char)flag_edge[tid];
  }
}

// init_pose [S,N,4] = x, y, heading, speed; size [S,N,2] = length, width
__global__ __launch_bounds__(256) void sim_init_kernel(int N, int E, const float* __restrict__ init_pose,
                                                       const float* __restrict__ size, const float* __restrict__ edges,
                                                       const unsigned char* __restrict__ exists,
                                                       float* __restrict__ phys, float* __restrict__ hist_states,
                                                       unsigned char* __restrict__ coll, int Tmax1) {
  __shared__ float corner[64][8];
  __shared__ float box[64][4];
  __shared__ int flag_veh[64], flag_edge[64];
  __shared__ float px[64], py[64], hd[64], sp[64];
  const int s = blockIdx.x, tid = threadIdx.x;
  if (tid < N) {
    const float* ip = init_pose + ((size_t)s * N + tid) * 4;
    float* p = phys + ((size_t)s * N + tid) * PHYS_STRIDE;
    const float L = size[((size_t)s * N + tid) * 2 + 0], Wd = size[((size_t)s * N + tid) * 2 + 1];
    const float heading = ip[2], speed = ip[3];
    local_center(Wd, L, &p[P_LCX], &p[P_LCY]);
    set_transform(p, 0.f, 0.f, (float)((double)heading - M_PI_D * 0.5f));   // vehicle.cc:168
    set_transform(p, ip[0], ip[1], p[P_A]);                                  // vehicle.cc:169
    p[P_VX] = speed * cosf(heading);
    p[P_VY] = speed * sinf(heading);
    p[P_W] = 0.f; p[P_SLEEP] = 0.f; p[P_AWAKE] = 1.f;
    p[P_THR] = 0.f; p[P_BRK] = 0.f; p[P_STEER] = 0.f;
    p[P_HEADING] = heading; p[P_SPEED] = speed; p[17] = p[18] = p[19] = 0.f;
    px[tid] = ip[0]; py[tid] = ip[1]; hd[tid] = heading; sp[tid] = speed;
  }
  __syncthreads();
  collide_and_record(s, N, E, size, edges, exists, hist_states, coll, 0, Tmax1, corner, box, flag_veh, flag_edge, px, py,
                     hd, sp);
}

// One rollout step t -> t+1.  Actions: token ids [S,N] (int32, action vocabulary index) or, if act_f64 != nullptr,
// explicit (accel, steer) doubles [S,N,2] (log-replay / facade path).
__global__ __launch_bounds__(256) void sim_step_kernel(int N, int E, const int* __restrict__ act_tok,
                                                       const double* __restrict__ act_f64, SimDiscretisation dz,
                                                       const float* __restrict__ size, const float* __restrict__ edges,
                                                       const unsigned char* __restrict__ exists,
                                                       float* __restrict__ phys, float* __restrict__ hist_states,
                                                       unsigned char* __restrict__ coll, double* __restrict__ applied,
                                                       int t, int Tmax1, float dt, int kinematic) {
  __shared__ float corner[64][8];
  __shared__ float box[64][4];
  __shared__ int flag_veh[64], flag_edge[64];
  __shared__ float px[64], py[64], hd[64], sp[64];
  const int s = blockIdx.x, tid = threadIdx.x;
  if (tid < N) {
    const size_t sn = (size_t)s * N + tid;
    float* p = phys + sn * PHYS_STRIDE;
    const float L = size[sn * 2 + 0];
    double accel, steer;
    if (!exists[sn]) {                                   // autoregressive_policy.py:260-263
      accel = 0.0; steer = 0.0;
      set_transform(p, -1000000.f, -1000000.f, p[P_A]);
    } else if (act_f64) {
      accel = act_f64[sn * 2 + 0];
      steer = act_f64[sn * 2 + 1];
    } else {                                             // dataset.py:322-338 undiscretize_actions (float64)
      const int tok = act_tok[sn];
      accel = (double)(tok / dz.n_steer) / (double)(dz.n_accel - 1);
      steer = (double)(tok % dz.n_steer) / (double)(dz.n_steer - 1);
      accel = accel * (dz.max_accel - dz.min_accel) + dz.min_accel;
      steer = steer * (dz.max_steer - dz.min_steer) + dz.min_steer;
    }
    if (applied) { applied[sn * 2 + 0] = accel; applied[sn * 2 + 1] = steer; }
    if (kinematic) {                                     // Object::KinematicBicycleStep, object.cc:126-137 (optional mode)
      const float kPi = 3.14159265358979323846f, kTwoPi = 2.0f * 3.14159265358979323846f;
      const float a = (float)accel, d = (float)steer;
      float heading = p[P_HEADING];
      float speed = p[P_SPEED];
      const float v = speed + 0.5f * a * dt;
      const float tan_delta = tanf(d);
      const float beta = atanf(0.5f * tan_delta);
      const float dx = v * cosf(heading + beta), dy = v * sinf(heading + beta);
      const float w = v * cosf(beta) * tan_delta / L;
      const float nx = p[P_PX] + dx * dt, ny = p[P_PY] + dy * dt;
      const float ang = fmodf(heading + w * dt, kTwoPi);
      heading = ang > kPi ? ang - kTwoPi : (ang < -kPi ? ang + kTwoPi : ang);
      speed = speed + a * dt;
      p[P_PX] = nx; p[P_PY] = ny; p[P_HEADING] = heading; p[P_SPEED] = speed;
      px[tid] = nx; py[tid] = ny; hd[tid] = heading; sp[tid] = speed;
    } else {
      // ---- setters (vehicle.cc:107-135 -> FreeCar.cpp:66-86)
      if (accel > 0.0) {
        const float a = (float)accel;
        p[P_THR] = (a > 0) ? 1.0f * a : 0.f * a;
        p[P_BRK] = 0.f;
      } else {
        const float bk = (float)fabs(accel);
        if (!((double)fabsf(bk) < 0.001)) { p[P_THR] = 0; p[P_BRK] = 1.0f * bk; }
      }
      p[P_STEER] = (float)steer;
      // ---- FreeCar::Step (FreeCar.cpp:98-186)
      const float thr = p[P_THR], brk = p[P_BRK], st = p[P_STEER];
      float target, acc;
      if (thr > 0.f) {
        if (thr > brk) { target = 50.f; acc = thr - brk; } else { target = 0.f; acc = brk - thr; }
      } else {
        if (thr < -brk) { target = -5.f; acc = -thr - brk; } else { target = 0.f; acc = brk + thr; }
      }
      float ang = p[P_W];
      const float beta = (float)atan(0.5 * (double)tanf(st));
      const float c = cosf(p[P_A] + beta), sn_ = sinf(p[P_A] + beta);
      const float fx = -sn_, fy = c, rx = c, ry = sn_;
      float sf = p[P_VX] * fx + p[P_VY] * fy;
      float sr = p[P_VX] * rx + p[P_VY] * ry;
      const float dv = acc * dt;
      if (sf < target) sf = fminf(sf + dv, target); else sf = fmaxf(sf - dv, target);
      float steer_w = 0.f;
      if (fabs((double)st) > 0.0000001) {
        const float ray = 1.f / tanf(st) * L / cosf(beta);
        steer_w = sf / ray;
      }
      sr = dampen(sr, 0, 25.f, dt);
      ang = dampen(ang, steer_w, 10.f, dt);
      const float nvx = rx * sr + fx * sf, nvy = ry * sr + fy * sf;
      float awake = p[P_AWAKE], sleep_t = p[P_SLEEP];
      if (nvx * nvx + nvy * nvy > 0.0f) { awake = 1.f; sleep_t = 0.f; }   // b2Body::SetLinearVelocity
      if (ang * ang > 0.0f) { awake = 1.f; sleep_t = 0.f; }               // b2Body::SetAngularVelocity
      float vx = nvx, vy = nvy, w = ang;
      // ---- b2Island::Solve (single-body island, no contacts)
      if (awake != 0.f) {
        const float tx = dt * vx, ty = dt * vy;
        if (tx * tx + ty * ty > B2_MAXTRANSLATION * B2_MAXTRANSLATION) {
          const float ratio = B2_MAXTRANSLATION / sqrtf(tx * tx + ty * ty);
          vx *= ratio; vy *= ratio;
        }
        const float rot = dt * w;
        if (rot * rot > B2_MAXROTATION * B2_MAXROTATION) {
          const float ratio = B2_MAXROTATION / fabsf(rot);
          w *= ratio;
        }
        p[P_CX] += dt * vx;
        p[P_CY] += dt * vy;
        p[P_A] += dt * w;
        if (w * w > B2_ANGSLEEPTOL * B2_ANGSLEEPTOL || vx * vx + vy * vy > B2_LINSLEEPTOL * B2_LINSLEEPTOL) sleep_t = 0.0f;
        else sleep_t += dt;
        const float qs = sinf(p[P_A]), qc = cosf(p[P_A]);                  // SynchronizeTransform
        p[P_PX] = p[P_CX] - (qc * p[P_LCX] - qs * p[P_LCY]);
        p[P_PY] = p[P_CY] - (qs * p[P_LCX] + qc * p[P_LCY]);
        if (sleep_t >= B2_TIMETOSLEEP) { awake = 0.f; sleep_t = 0.f; vx = vy = 0.f; w = 0.f; }
      }
      p[P_VX] = vx; p[P_VY] = vy; p[P_W] = w; p[P_AWAKE] = awake; p[P_SLEEP] = sleep_t;
      // ---- Vehicle::Step read-back (vehicle.cc:45-55)
      px[tid] = p[P_PX];
      py[tid] = p[P_PY];
      sp[tid] = sqrtf(vx * vx + vy * vy);
      hd[tid] = (float)((double)p[P_A] + M_PI_D * 0.5f);
      p[P_HEADING] = hd[tid]; p[P_SPEED] = sp[tid];
    }
  }
  __syncthreads();
  collide_and_record(s, N, E, size, edges, exists, hist_states, coll, t + 1, Tmax1, corner, box, flag_veh, flag_edge, px,
                     py, hd, sp);
}

int launch_sim_init(int S, int N, int E, const float* init_pose, const float* size, const float* edges,
                    const unsigned char* exists, float* phys, float* hist_states, unsigned char* coll, int Tmax1,
                    hipStream_t st) {
  if (S <= 0) return CTRLSIM_OK;
  if (N < 1 || N > 64 || E < 0) return CTRLSIM_EINVAL;
  hipLaunchKernelGGL(sim_init_kernel, dim3(S), dim3(256), 0, st, N, E, init_pose, size, edges, exists, phys, hist_states,
                     coll, Tmax1);
  return ctrlsim_launch_status();
}

int launch_sim_step(int S, int N, int E, const int* act_tok, const double* act_f64, const double* disc6,
                    const float* size, const float* edges, const unsigned char* exists, float* phys,
                    float* hist_states, unsigned char* coll, double* applied, int t, int Tmax1, float dt, int kinematic,
                    hipStream_t st) {
  if (S <= 0) return CTRLSIM_OK;
  if (N < 1 || N > 64 || E < 0 || t < 0 || t + 1 >= Tmax1 || (!act_tok && !act_f64)) return CTRLSIM_EINVAL;
  SimDiscretisation dz{disc6[0], disc6[1], disc6[2], disc6[3], (int)disc6[4], (int)disc6[5]};
  hipLaunchKernelGGL(sim_step_kernel, dim3(S), dim3(256), 0, st, N, E, act_tok, act_f64, dz, size, edges, exists, phys,
                     hist_states, coll, applied, t, Tmax1, dt, kinematic);
  return ctrlsim_launch_status();
}
