// Batched per-step vehicle update + collision flags for S independent scenarios (gfx950, float32).
// Compiled with -ffp-contract=off: results must equal the reference's plain IEEE float32 evaluation.
//
// What one launch replaces in the reference (paths relative to /root/reference):
//   policies/autoregressive_policy.py:256-274  act(): token -> (accel, steer), throttle / brake / steer setters,
//                                              teleport of non-existing vehicles to (-1e6,-1e6)
//   nocturne/cpp/src/vehicle.cc:75-135         setters -> FreeCar::Throttle/Brake/Turn
//   nocturne/cpp/src/physics/FreeCar.cpp:66-186 FreeCar controller (DampenSpeed, slip angle beta, steering radius)
//   third_party/box2d/src/dynamics/b2_island.cpp:194-229,279-310,349-392  integrate, translation/rotation clamps
//                                              (patched b2_maxTranslation = 5.0f, b2_common.h:95), auto-sleep
//   third_party/box2d/src/collision/b2_collide_polygon.cpp, b2_collision.cpp; src/dynamics/b2_contact.cpp:165-245,
//   b2_world.cpp:393-560, b2_contact_solver.cpp          box-box contacts between vehicles: manifolds, impulse matching,
//                                              islands, warm-started velocity solver with the 2-point block solver,
//                                              position solver (when a contact-state buffer is given; see "contacts" below)
//   nocturne/cpp/src/vehicle.cc:45-55          read-back: position <- xf.p, speed <- |v|, heading <- angle + pi/2
//   nocturne/cpp/src/scenario.cc:266-328       reset flags, vehicle-vehicle SAT and vehicle-road-edge tests with the
//                                              strict-AABB candidate predicate (aabb.h:47-50, bvh.h:181-193)
//   policies/policy.py:68-79                   history append of the 8-float state row
// One workgroup per scenario; lane i < N owns vehicle i (N <= 64: one wavefront of vehicles), all 256 threads share
// the N x N and N x E collision tests.  State is SoA-per-scenario [S, N, ...] so a wave's loads are contiguous.
#include <cstdlib>
#include "common.h"
// -DSIM_JITTER (tools only): random per-wave stalls around every workgroup barrier of the step — a missing barrier / a data race
// between the waves of a scenario's workgroup then shows as a run-to-run difference even with nothing else on the device
#ifdef SIM_JITTER
__device__ __forceinline__ void sim_jitter(int point) {
  unsigned h = ((unsigned)__builtin_amdgcn_s_memtime() ^ ((threadIdx.x >> 6) * 0x9E3779B9u + point * 0x85EBCA6Bu)) * 2654435761u;
  h = __builtin_amdgcn_readfirstlane(h);
  if ((h >> 29) < 3) {
    const int n = (h >> 20) & 15;
    for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(100);
  }
}
#define SYNCJ() do { sim_jitter(__LINE__); __syncthreads(); sim_jitter(__LINE__ + 4096); } while (0)
#elif defined(SIM_FENCE)
// (tools only) every workgroup barrier also releases / acquires global memory at agent scope (L2 write-back, L1 invalidate): if the
// co-residency hazard of DESIGN.md section 4 travels through the vector-memory caches, this build does not show it
#define SYNCJ() do { __threadfence(); __syncthreads(); __threadfence(); } while (0)
#else
#define SYNCJ() __syncthreads()
#endif


#define PHYS_STRIDE 20
enum { P_CX = 0, P_CY, P_A, P_VX, P_VY, P_W, P_SLEEP, P_AWAKE, P_THR, P_BRK, P_STEER, P_LCX, P_LCY, P_PX, P_PY,
       P_HEADING, P_SPEED,     // P_HEADING/P_SPEED: Object-level heading_/speed_ (what Python reads)
       P_TELE, P_TX, P_TY };   // a pending Vehicle::set_position(x, y) (ctrlsim_sim_set_position), applied by the next step

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
__device__ void local_center(float width, float length, float* lcx, float* lcy, float* inv_mass_out = nullptr,
                             float* inv_i_out = nullptr) {
  const float hx = width / 2, hy = length / 2;
  const float vx[4] = {-hx, hx, hx, -hx}, vy[4] = {-hy, -hy, hy, hy};
  float cx = 0.0f, cy = 0.0f, area = 0.0f, I = 0.0f;
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
    const float intx2 = e1x * e1x + e2x * e1x + e2x * e2x;
    const float inty2 = e1y * e1y + e2y * e1y + e2y * e2y;
    I += (0.25f * k_inv3 * D) * (intx2 + inty2);
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
  if (inv_mass_out) {                                   // b2PolygonShape::ComputeMass inertia + b2Body::ResetMassData
    float mI = 20.f * I;
    mI += mass * ((mcx * mcx + mcy * mcy) - (cx * cx + cy * cy));
    const float bI = mI - mass * (*lcx * *lcx + *lcy * *lcy);
    *inv_mass_out = inv_mass;
    *inv_i_out = 1.0f / bI;
  }
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

#ifdef SIM_TIMING
__device__ unsigned long long g_sim_t[12];
extern "C" void ctrlsim_sim_timing(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sim_t), sizeof(g_sim_t));
  if (reset) { unsigned long long z[12] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_sim_t), z, sizeof(z)); }
}
#define SIMT(k) if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_sim_t[k], n_ - tl_); tl_ = n_; }
#else
#define SIMT(k)
#endif
#define GRID 16               // broad-phase grid of the vehicle x road-edge tests (GRID x GRID cells, a 64-bit vehicle mask each)
// shared by init and step: corners/AABBs into LDS, flags, history row
__device__ void collide_and_record(int s, int N, int E, const float* __restrict__ size, const float* __restrict__ edges,
                                   const unsigned char* __restrict__ exists, float* __restrict__ hist_states,
                                   unsigned char* __restrict__ coll, int t_row, int Tmax1, float (*corner)[8],
                                   float (*box)[4], int* flag_veh, int* flag_edge, const float* px, const float* py,
                                   const float* heading, const float* speed) {
  const int tid = threadIdx.x;
#ifdef SIM_TIMING
  unsigned long long tl_ = __builtin_amdgcn_s_memtime();
#endif
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
  SYNCJ();
  SIMT(10)
  for (int p = tid; p < N * N; p += blockDim.x) {
    const int i = p / N, j = p - i * N;
    if (i == j) continue;
    if (!(box[i][0] < box[j][2] && box[i][2] > box[j][0] && box[i][1] < box[j][3] && box[i][3] > box[j][1])) continue;
    if (box_box(corner[i], corner[j])) atomicOr(&flag_veh[i], 1);
  }
  SIMT(11)
  // Vehicle x road-edge segment tests behind a broad phase.  The reference tests every vehicle against every segment
  // (bounding-box pre-test, then the exact test); hits are a few dozen of ~800 000 pairs.  Here the vehicles' bounding boxes are
  // binned into a GRID x GRID grid over the area the existing vehicles occupy (a 64-bit vehicle mask per cell), a thread per
  // segment ORs the masks of the cells its bounding box touches and runs the reference's two tests on the vehicles of that mask
  // only.  Conservative: boxes that overlap share a point, that point lies inside a vehicle box and therefore inside the grid,
  // and both sides map it to the same cell (same expression, clamped) — so no pair the exhaustive sweep would flag is skipped.
  // (The exhaustive loop was 40 % of the kernel: ~500 cycles per pair on dependent LDS reads of the boxes.)
  const float* eg = edges + (size_t)s * E * 4;
  __shared__ unsigned cell_lo[GRID * GRID], cell_hi[GRID * GRID];
  __shared__ float gbox[4];
  __shared__ int gany;
  __shared__ unsigned live_lo, live_hi;                        // existing vehicles (the ones the grid bins)
  for (int c = tid; c < GRID * GRID; c += blockDim.x) { cell_lo[c] = 0u; cell_hi[c] = 0u; }
  if (tid < 64) {                                              // wave 0: bounding box of the existing vehicles
    const bool live = tid < N && exists[(size_t)s * N + tid];
    float m0 = live ? box[tid][0] : 3.402823466e+38f, m1 = live ? box[tid][1] : 3.402823466e+38f;
    float m2 = live ? box[tid][2] : -3.402823466e+38f, m3 = live ? box[tid][3] : -3.402823466e+38f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      m0 = fminf(m0, __shfl_xor(m0, o)); m1 = fminf(m1, __shfl_xor(m1, o));
      m2 = fmaxf(m2, __shfl_xor(m2, o)); m3 = fmaxf(m3, __shfl_xor(m3, o));
    }
    const unsigned long long lm = __ballot(live);
    if (tid == 0) {
      gbox[0] = m0; gbox[1] = m1; gbox[2] = m2; gbox[3] = m3; gany = m2 >= m0 ? 1 : 0;
      live_lo = (unsigned)lm; live_hi = (unsigned)(lm >> 32);
    }
  }
  SYNCJ();
  if (gany) {
    const float gx0 = gbox[0], gy0 = gbox[1];
    const float ihx = (float)GRID / fmaxf(gbox[2] - gx0, 1e-3f), ihy = (float)GRID / fmaxf(gbox[3] - gy0, 1e-3f);
    auto cellx = [&](float x) { return min(GRID - 1, max(0, (int)floorf((x - gx0) * ihx))); };
    auto celly = [&](float y) { return min(GRID - 1, max(0, (int)floorf((y - gy0) * ihy))); };
    if (tid < N && exists[(size_t)s * N + tid]) {
      const int x0 = cellx(box[tid][0]), x1 = cellx(box[tid][2]), y0 = celly(box[tid][1]), y1 = celly(box[tid][3]);
      for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
          if (tid < 32) atomicOr(&cell_lo[y * GRID + x], 1u << tid);
          else atomicOr(&cell_hi[y * GRID + x], 1u << (tid - 32));
        }
    }
    SYNCJ();
    for (int e = tid; e < E; e += blockDim.x) {
      const f32x4 sg = *reinterpret_cast<const f32x4*>(eg + (size_t)e * 4);
      const float s0 = fminf(sg[0], sg[2]), s1 = fminf(sg[1], sg[3]), s2 = fmaxf(sg[0], sg[2]), s3 = fmaxf(sg[1], sg[3]);
      if (!(s2 > gbox[0] && s0 < gbox[2] && s3 > gbox[1] && s1 < gbox[3])) continue;     // misses every existing vehicle's box
      const int x0 = cellx(s0), x1 = cellx(s2), y0 = celly(s1), y1 = celly(s3);
      unsigned lo = 0u, hi = 0u;
      if ((x1 - x0 + 1) * (y1 - y0 + 1) > 16) { lo = live_lo; hi = live_hi; }             // a very long segment: every binned vehicle
      else
        for (int y = y0; y <= y1; ++y)
          for (int x = x0; x <= x1; ++x) { lo |= cell_lo[y * GRID + x]; hi |= cell_hi[y * GRID + x]; }
      unsigned long long m = ((unsigned long long)hi << 32) | lo;
      while (m) {
        const int i = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (i >= N) break;
        if (!(box[i][0] < s2 && box[i][2] > s0 && box[i][1] < s3 && box[i][3] > s1)) continue;
        if (box_seg(corner[i], sg[0], sg[1], sg[2], sg[3])) atomicOr(&flag_edge[i], 1);
      }
    }
  }
  SYNCJ();
  if (tid < N) {
    unsigned char* c = coll + (((size_t)s * N + tid) * Tmax1 + t_row) * 2;
    c[0] = (unsigned char)flag_veh[tid];
    c[1] = (unsigned char)flag_edge[tid];
  }
}

// ---------------------------------------------------------------------------------------------------- contacts
// Box2D's treatment of overlapping vehicle boxes, restated from the C++ sources cited at the top (and, line for line, from
// this repo's CPU oracle oracle/sim_oracle.c, which is pinned bit-exactly against the real Box2D for two-body islands).
// Per scenario and step:  (1) every vehicle pair: b2CollidePolygons manifold + impulse matching by feature id (parallel
// over pairs), (2) islands by depth-first search over touching pairs, (3) per island the sequential-impulse solver:
// warm start, 8 velocity iterations (friction, then normal / 2-point block solver), integration with the translation and
// rotation clamps, 3 position iterations, sleep.  Islands of one body integrate on their own lanes in parallel; islands
// with contacts are solved by lane 0 in Box2D's sequential order (Gauss-Seidel is order dependent).
// Contact state per pair (i < j), CS_STRIDE floats: two manifold points {local x, y, normal impulse, tangent impulse, id},
// local normal, local point, type, point count, touching flag.
#define CS_STRIDE 20
#define CS_LN 10
#define CS_LP 12
#define CS_TYPE 14
#define CS_COUNT 15
#define CS_TOUCH 16
#define CS_EXISTS 17     // contact exists (fat AABBs overlap since it was created)
#define CS_STAMP 18      // creation order (larger = newer): decides the contact order inside islands
// per-scenario tail after the NP pair records: [inv_dt0, stamp counter, new-contacts flag, move-buffer length],
// fat AABB per proxy [N][4], moved flag [N], sweep.c0 / a0 [N][3], move buffer [4N], b2DynamicTree nodes [2N][8]
#define CS_TAIL 6        // m_inv_dt0, contact stamp, m_newContacts, move count, tree root, tree free-list head
#define TN_STRIDE 8      // tree node: box lower x, y, upper x, y, parent (= next in the free list), child1, child2, height
#define CS_PER(N) ((N) * ((N) - 1) / 2 * CS_STRIDE + CS_TAIL + 12 * (N) + 2 * (N) * TN_STRIDE)
#define MAX_ISLAND_CONTACTS 192   // >= 3 N - 6 = 186 at N = 64: the touching graph of disjoint convex boxes is planar
#define B2_AABB_EXT 0.1f
#define B2_AABB_MULT 4.0f
#define B2_LINEAR_SLOP 0.005f
#define B2_POLY_RADIUS (2.0f * B2_LINEAR_SLOP)
#define B2_FLT_MAX 3.402823466e+38F
#define B2_FLT_EPS 1.1920928955078125e-7f

struct V2 { float x, y; };
struct Rot { float s, c; };
struct Xf { V2 p; Rot q; };
__device__ __forceinline__ float b2maxf(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float b2minf(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ V2 rot_mul(Rot q, V2 v) { return v2(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
__device__ __forceinline__ V2 rot_mulT(Rot q, V2 v) { return v2(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
__device__ __forceinline__ V2 xf_mul(Xf t, V2 v) {
  return v2((t.q.c * v.x - t.q.s * v.y) + t.p.x, (t.q.s * v.x + t.q.c * v.y) + t.p.y);
}
__device__ __forceinline__ V2 xf_mulT(Xf t, V2 v) {
  const float px = v.x - t.p.x, py = v.y - t.p.y;
  return v2(t.q.c * px + t.q.s * py, -t.q.s * px + t.q.c * py);
}
__device__ __forceinline__ Xf xf_mulT_xf(Xf A, Xf B) {
  Xf C;
  C.q.s = A.q.c * B.q.s - A.q.s * B.q.c;
  C.q.c = A.q.c * B.q.c + A.q.s * B.q.s;
  C.p = rot_mulT(A.q, v2(B.p.x - A.p.x, B.p.y - A.p.y));
  return C;
}
__device__ __forceinline__ float dot2(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float crossvv(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ V2 cross_vs(V2 a, float s) { return v2(s * a.y, -s * a.x); }
__device__ __forceinline__ V2 cross_sv(float s, V2 a) { return v2(-s * a.y, s * a.x); }

struct Box { V2 v[4], n[4]; };
__device__ __forceinline__ Box box_of(float width, float length) {   // SetAsBox(width/2, length/2), FreeCar.cpp:39
  const float hx = width / 2, hy = length / 2;
  Box b;
  b.v[0] = v2(-hx, -hy); b.v[1] = v2(hx, -hy); b.v[2] = v2(hx, hy); b.v[3] = v2(-hx, hy);
  b.n[0] = v2(0.0f, -1.0f); b.n[1] = v2(1.0f, 0.0f); b.n[2] = v2(0.0f, 1.0f); b.n[3] = v2(-1.0f, 0.0f);
  return b;
}
struct ClipV { V2 v; unsigned ia, ib, ta, tb; };

// Vertices / normals of SetAsBox(hx, hy) by index, without arrays: collide_boxes selects its reference / incident polygon and
// their edges at run time, and a run-time index into a local array (or a reference chosen by `flip`) would put the polygons into
// scratch memory (240 bytes per lane, written and re-read per contact).  This kernel keeps nothing there (tests/test_isa_checks.py).
struct HBox { float hx, hy; };
__device__ __forceinline__ HBox hbox_of(const Box& b) { HBox h; h.hx = b.v[2].x; h.hy = b.v[2].y; return h; }
__device__ __forceinline__ V2 hb_v(HBox b, int i) { return v2((i == 1 || i == 2) ? b.hx : -b.hx, i >= 2 ? b.hy : -b.hy); }
__device__ __forceinline__ V2 hb_n(int i) { return v2(i == 1 ? 1.0f : (i == 3 ? -1.0f : 0.0f), i == 0 ? -1.0f : (i == 2 ? 1.0f : 0.0f)); }

__device__ float find_max_separation(int* edge, HBox p1, Xf xf1, HBox p2, Xf xf2) {
  const Xf xf = xf_mulT_xf(xf2, xf1);
  int best = 0;
  float max_sep = -B2_FLT_MAX;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const V2 n = rot_mul(xf.q, hb_n(i));
    const V2 v1 = xf_mul(xf, hb_v(p1, i));
    float si = B2_FLT_MAX;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const V2 pj = hb_v(p2, j);
      const float sij = dot2(n, v2(pj.x - v1.x, pj.y - v1.y));
      if (sij < si) si = sij;
    }
    if (si > max_sep) { max_sep = si; best = i; }
  }
  *edge = best;
  return max_sep;
}
// b2ClipSegmentToLine for the only outcome the caller uses (two output points; anything else makes it return 0 points):
//   both inputs behind the plane -> (in0, in1);  in0 behind, in1 in front -> (in0, intersection);  in1 behind, in0 in front ->
//   (in1, intersection) — the order in which the reference appends them.  No arrays: see hb_v.
__device__ __forceinline__ bool clip_segment2(ClipV& o0, ClipV& o1, const ClipV& i0, const ClipV& i1, V2 normal, float offset,
                                              int vertex_index_a) {
  const float d0 = dot2(normal, i0.v) - offset;
  const float d1 = dot2(normal, i1.v) - offset;
  const bool b0 = d0 <= 0.0f, b1 = d1 <= 0.0f, cross = d0 * d1 < 0.0f;
  const int count = (b0 ? 1 : 0) + (b1 ? 1 : 0) + (cross ? 1 : 0);
  if (count != 2) return false;                    // (three is impossible: both behind excludes a crossing)
  ClipV x;
  {
    const float interp = d0 / (d0 - d1);
    x.v = v2(i0.v.x + interp * (i1.v.x - i0.v.x), i0.v.y + interp * (i1.v.y - i0.v.y));
    x.ia = (unsigned)vertex_index_a;
    x.ib = i0.ib;
    x.ta = 0;      // e_vertex
    x.tb = 1;      // e_face
  }
  o0 = b0 ? i0 : i1;
  o1 = (b0 && b1) ? i1 : x;
  return true;
}
// b2CollidePolygons into the contact record m (point impulses are set by the caller); returns the point count
__device__ int collide_boxes(float* m, const Box& A_, Xf xfA, const Box& B_, Xf xfB) {
  const HBox A = hbox_of(A_), B = hbox_of(B_);
  const float total_radius = B2_POLY_RADIUS + B2_POLY_RADIUS;
  int edgeA = 0, edgeB = 0;
  const float sepA = find_max_separation(&edgeA, A, xfA, B, xfB);
  if (sepA > total_radius) return 0;
  const float sepB = find_max_separation(&edgeB, B, xfB, A, xfA);
  if (sepB > total_radius) return 0;
  const float k_tol = 0.1f * B2_LINEAR_SLOP;
  const bool flip = sepB > sepA + k_tol;
  HBox p1, p2;
  p1.hx = flip ? B.hx : A.hx; p1.hy = flip ? B.hy : A.hy;
  p2.hx = flip ? A.hx : B.hx; p2.hy = flip ? A.hy : B.hy;
  const Xf xf1 = flip ? xfB : xfA, xf2 = flip ? xfA : xfB;
  const int edge1 = flip ? edgeB : edgeA;
  ClipV inc0, inc1;
  {
    const V2 normal1 = rot_mulT(xf2.q, rot_mul(xf1.q, hb_n(edge1)));
    int index = 0;
    float min_dot = B2_FLT_MAX;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = dot2(normal1, hb_n(i));
      if (d < min_dot) { min_dot = d; index = i; }
    }
    const int i1 = index, i2 = i1 + 1 < 4 ? i1 + 1 : 0;
    inc0.v = xf_mul(xf2, hb_v(p2, i1)); inc0.ia = (unsigned)edge1; inc0.ib = (unsigned)i1; inc0.ta = 1; inc0.tb = 0;
    inc1.v = xf_mul(xf2, hb_v(p2, i2)); inc1.ia = (unsigned)edge1; inc1.ib = (unsigned)i2; inc1.ta = 1; inc1.tb = 0;
  }
  const int iv1 = edge1, iv2 = edge1 + 1 < 4 ? edge1 + 1 : 0;
  V2 v11 = hb_v(p1, iv1), v12 = hb_v(p1, iv2);
  V2 lt = v2(v12.x - v11.x, v12.y - v11.y);
  {
    const float len = sqrtf(lt.x * lt.x + lt.y * lt.y);
    if (!(len < B2_FLT_EPS)) { const float inv = 1.0f / len; lt.x *= inv; lt.y *= inv; }
  }
  const V2 local_normal = cross_vs(lt, 1.0f);
  const V2 plane_point = v2(0.5f * (v11.x + v12.x), 0.5f * (v11.y + v12.y));
  const V2 tangent = rot_mul(xf1.q, lt);
  const V2 normal = cross_vs(tangent, 1.0f);
  v11 = xf_mul(xf1, v11);
  v12 = xf_mul(xf1, v12);
  const float front_offset = dot2(normal, v11);
  const float side1 = -dot2(tangent, v11) + total_radius;
  const float side2 = dot2(tangent, v12) + total_radius;
  ClipV c10, c11, c20, c21;
  if (!clip_segment2(c10, c11, inc0, inc1, v2(-tangent.x, -tangent.y), side1, iv1)) return 0;
  if (!clip_segment2(c20, c21, c10, c11, tangent, side2, iv2)) return 0;
  m[CS_LN] = local_normal.x; m[CS_LN + 1] = local_normal.y;
  m[CS_LP] = plane_point.x; m[CS_LP + 1] = plane_point.y;
  m[CS_TYPE] = flip ? 2.0f : 1.0f;                         // e_faceB : e_faceA
  int pc = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const ClipV c = i == 0 ? c20 : c21;
    const float separation = dot2(normal, c.v) - front_offset;
    if (separation <= total_radius) {
      const V2 lp = xf_mulT(xf2, c.v);
      m[5 * pc + 0] = lp.x; m[5 * pc + 1] = lp.y;
      unsigned ia = c.ia, ib = c.ib, ta = c.ta, tb = c.tb;
      if (flip) { unsigned t = ia; ia = ib; ib = t; t = ta; ta = tb; tb = t; }
      m[5 * pc + 4] = __uint_as_float(ia | (ib << 8) | (ta << 16) | (tb << 24));
      ++pc;
    }
  }
  return pc;
}

// ---- broad phase: what fixes the ORDER in which Box2D creates contacts (b2_dynamic_tree.cpp:107-195
// MoveProxy fat AABBs, b2_broad_phase.cpp / .h BufferMove, UpdatePairs, QueryCallback; b2_fixture.cpp:156-178)
__device__ void shape_aabb(const Box& b, Xf xf, float* bb) {      // b2PolygonShape::ComputeAABB
  V2 lo = xf_mul(xf, b.v[0]), hi = lo;
  for (int i = 1; i < 4; ++i) {
    const V2 p = xf_mul(xf, b.v[i]);
    lo = v2(b2minf(lo.x, p.x), b2minf(lo.y, p.y));
    hi = v2(b2maxf(hi.x, p.x), b2maxf(hi.y, p.y));
  }
  bb[0] = lo.x - B2_POLY_RADIUS; bb[1] = lo.y - B2_POLY_RADIUS; bb[2] = hi.x + B2_POLY_RADIUS; bb[3] = hi.y + B2_POLY_RADIUS;
}
__device__ __forceinline__ bool aabb_contains(const float* a, const float* b) {
  return a[0] <= b[0] && a[1] <= b[1] && b[2] <= a[2] && b[3] <= a[3];
}
__device__ __forceinline__ bool aabb_overlap(const float* a, const float* b) {
  const float d1x = b[0] - a[2], d1y = b[1] - a[3], d2x = a[0] - b[2], d2y = a[1] - b[3];
  if (d1x > 0.0f || d1y > 0.0f) return false;
  if (d2x > 0.0f || d2y > 0.0f) return false;
  return true;
}
// b2Fixture::Synchronize + b2DynamicTree::MoveProxy for one proxy; returns true when the proxy was re-inserted (moved)
__device__ bool synchronize_fixture(float* fat, const Box& b, Xf xf1, Xf xf2) {
  float a1[4], a2[4], c[4];
  shape_aabb(b, xf1, a1); shape_aabb(b, xf2, a2);
  c[0] = b2minf(a1[0], a2[0]); c[1] = b2minf(a1[1], a2[1]); c[2] = b2maxf(a1[2], a2[2]); c[3] = b2maxf(a1[3], a2[3]);
  const float dx = 0.5f * (a2[0] + a2[2]) - 0.5f * (a1[0] + a1[2]);
  const float dy = 0.5f * (a2[1] + a2[3]) - 0.5f * (a1[1] + a1[3]);
  float fatn[4] = {c[0] - B2_AABB_EXT, c[1] - B2_AABB_EXT, c[2] + B2_AABB_EXT, c[3] + B2_AABB_EXT};
  const float ddx = B2_AABB_MULT * dx, ddy = B2_AABB_MULT * dy;
  if (ddx < 0.0f) fatn[0] += ddx; else fatn[2] += ddx;
  if (ddy < 0.0f) fatn[1] += ddy; else fatn[3] += ddy;
  if (aabb_contains(fat, c)) {
    const float huge[4] = {fatn[0] - 4.0f * B2_AABB_EXT, fatn[1] - 4.0f * B2_AABB_EXT, fatn[2] + 4.0f * B2_AABB_EXT,
                           fatn[3] + 4.0f * B2_AABB_EXT};
    if (aabb_contains(huge, fat)) return false;
  }
  fat[0] = fatn[0]; fat[1] = fatn[1]; fat[2] = fatn[2]; fat[3] = fatn[3];
  return true;
}
// ---- b2DynamicTree (b2_dynamic_tree.cpp:57-105 pool, :198-332 InsertLeaf, :334-393 RemoveLeaf, :397-534 Balance;
// b2_dynamic_tree.h:187-220 Query).  Its shape fixes the order in which one query reports its hits, i.e. the order in which
// contacts that begin in the same step are created.  One lane works on it.  Every vehicle takes a fresh leaf at creation and
// internal nodes recycle among themselves, so leaf(vehicle i) = i ? 2 i - 1 : 0 and ids compare like vehicle indices.
struct Tree {
  float* n;        // nodes [2N][TN_STRIDE]
  float* rf;       // rf[0] = root, rf[1] = free-list head
  __device__ float* bb(int i) const { return n + i * TN_STRIDE; }
  __device__ int parent(int i) const { return (int)n[i * TN_STRIDE + 4]; }
  __device__ int c1(int i) const { return (int)n[i * TN_STRIDE + 5]; }
  __device__ int c2(int i) const { return (int)n[i * TN_STRIDE + 6]; }
  __device__ int height(int i) const { return (int)n[i * TN_STRIDE + 7]; }
  __device__ void set_parent(int i, int v) { n[i * TN_STRIDE + 4] = (float)v; }
  __device__ void set_c1(int i, int v) { n[i * TN_STRIDE + 5] = (float)v; }
  __device__ void set_c2(int i, int v) { n[i * TN_STRIDE + 6] = (float)v; }
  __device__ void set_height(int i, int v) { n[i * TN_STRIDE + 7] = (float)v; }
  __device__ bool leaf(int i) const { return c1(i) < 0; }
  __device__ int root() const { return (int)rf[0]; }
  __device__ void set_root(int v) { rf[0] = (float)v; }
};
__device__ __forceinline__ int tree_leaf_of(int veh) { return veh ? 2 * veh - 1 : 0; }
__device__ __forceinline__ int tree_veh_of(int leaf) { return leaf ? (leaf + 1) >> 1 : 0; }
__device__ __forceinline__ void bb_combine(float* o, const float* a, const float* b) {
  const float x0 = b2minf(a[0], b[0]), y0 = b2minf(a[1], b[1]), x1 = b2maxf(a[2], b[2]), y1 = b2maxf(a[3], b[3]);
  o[0] = x0; o[1] = y0; o[2] = x1; o[3] = y1;
}
__device__ __forceinline__ float bb_perimeter(const float* a) {
  const float wx = a[2] - a[0], wy = a[3] - a[1];
  return 2.0f * (wx + wy);
}
__device__ void tree_reset(Tree T, int N) {            // empty tree, free list 0 -> 1 -> ... (b2DynamicTree ctor / pool growth)
  for (int i = 0; i < 2 * N; ++i) { T.set_parent(i, i + 1 < 2 * N ? i + 1 : -1); T.set_c1(i, -1); T.set_c2(i, -1); T.set_height(i, -1); }
  T.rf[0] = -1.f; T.rf[1] = 0.f;
}
__device__ int tree_alloc(Tree T) {
  const int id = (int)T.rf[1];
  T.rf[1] = (float)T.parent(id);
  T.set_parent(id, -1); T.set_c1(id, -1); T.set_c2(id, -1); T.set_height(id, 0);
  return id;
}
__device__ void tree_free(Tree T, int id) { T.set_parent(id, (int)T.rf[1]); T.set_height(id, -1); T.rf[1] = (float)id; }
__device__ int tree_balance(Tree T, int iA) {
  if (T.leaf(iA) || T.height(iA) < 2) return iA;
  const int iB = T.c1(iA), iC = T.c2(iA);
  const int balance = T.height(iC) - T.height(iB);
  auto hmax = [&](int a, int b) { const int ha = T.height(a), hb = T.height(b); return 1 + (ha > hb ? ha : hb); };
  if (balance > 1) {                                   // rotate C up
    const int iF = T.c1(iC), iG = T.c2(iC);
    T.set_c1(iC, iA); T.set_parent(iC, T.parent(iA)); T.set_parent(iA, iC);
    const int pc = T.parent(iC);
    if (pc >= 0) { if (T.c1(pc) == iA) T.set_c1(pc, iC); else T.set_c2(pc, iC); } else T.set_root(iC);
    const bool f_up = T.height(iF) > T.height(iG);
    const int up = f_up ? iF : iG, down = f_up ? iG : iF;
    T.set_c2(iC, up); T.set_c2(iA, down); T.set_parent(down, iA);
    bb_combine(T.bb(iA), T.bb(iB), T.bb(down)); bb_combine(T.bb(iC), T.bb(iA), T.bb(up));
    T.set_height(iA, hmax(iB, down)); T.set_height(iC, hmax(iA, up));
    return iC;
  }
  if (balance < -1) {                                  // rotate B up
    const int iD = T.c1(iB), iE = T.c2(iB);
    T.set_c1(iB, iA); T.set_parent(iB, T.parent(iA)); T.set_parent(iA, iB);
    const int pb = T.parent(iB);
    if (pb >= 0) { if (T.c1(pb) == iA) T.set_c1(pb, iB); else T.set_c2(pb, iB); } else T.set_root(iB);
    const bool d_up = T.height(iD) > T.height(iE);
    const int up = d_up ? iD : iE, down = d_up ? iE : iD;
    T.set_c2(iB, up); T.set_c1(iA, down); T.set_parent(down, iA);
    bb_combine(T.bb(iA), T.bb(iC), T.bb(down)); bb_combine(T.bb(iB), T.bb(iA), T.bb(up));
    T.set_height(iA, hmax(iC, down)); T.set_height(iB, hmax(iA, up));
    return iB;
  }
  return iA;
}
__device__ void tree_refit_up(Tree T, int index) {       // walk to the root: rebalance, refit heights and boxes
  while (index >= 0) {
    index = tree_balance(T, index);
    const int a = T.c1(index), b = T.c2(index);
    const int ha = T.height(a), hb = T.height(b);
    T.set_height(index, 1 + (ha > hb ? ha : hb));
    bb_combine(T.bb(index), T.bb(a), T.bb(b));
    index = T.parent(index);
  }
}
__device__ void tree_insert_leaf(Tree T, int leaf) {
  if (T.root() < 0) { T.set_root(leaf); T.set_parent(leaf, -1); return; }
  float lb[4] = {T.bb(leaf)[0], T.bb(leaf)[1], T.bb(leaf)[2], T.bb(leaf)[3]};
  int index = T.root();
  while (!T.leaf(index)) {                             // surface-area heuristic descent
    const int a = T.c1(index), b = T.c2(index);
    const float area = bb_perimeter(T.bb(index));
    float comb[4]; bb_combine(comb, T.bb(index), lb);
    const float combined = bb_perimeter(comb);
    const float cost = 2.0f * combined;
    const float inherit = 2.0f * (combined - area);
    float t[4], cost1, cost2;
    bb_combine(t, lb, T.bb(a));
    if (T.leaf(a)) cost1 = bb_perimeter(t) + inherit;
    else { const float old_area = bb_perimeter(T.bb(a)), new_area = bb_perimeter(t); cost1 = (new_area - old_area) + inherit; }
    bb_combine(t, lb, T.bb(b));
    if (T.leaf(b)) cost2 = bb_perimeter(t) + inherit;
    else { const float old_area = bb_perimeter(T.bb(b)), new_area = bb_perimeter(t); cost2 = new_area - old_area + inherit; }
    if (cost < cost1 && cost < cost2) break;
    index = cost1 < cost2 ? a : b;
  }
  const int sibling = index, old_parent = T.parent(sibling);
  const int np = tree_alloc(T);
  T.set_parent(np, old_parent);
  bb_combine(T.bb(np), lb, T.bb(sibling));
  T.set_height(np, T.height(sibling) + 1);
  if (old_parent >= 0) { if (T.c1(old_parent) == sibling) T.set_c1(old_parent, np); else T.set_c2(old_parent, np); }
  else T.set_root(np);
  T.set_c1(np, sibling); T.set_c2(np, leaf);
  T.set_parent(sibling, np); T.set_parent(leaf, np);
  tree_refit_up(T, np);
}
__device__ void tree_remove_leaf(Tree T, int leaf) {
  if (leaf == T.root()) { T.set_root(-1); return; }
  const int parent = T.parent(leaf), grand = T.parent(parent);
  const int sibling = T.c1(parent) == leaf ? T.c2(parent) : T.c1(parent);
  if (grand >= 0) {
    if (T.c1(grand) == parent) T.set_c1(grand, sibling); else T.set_c2(grand, sibling);
    T.set_parent(sibling, grand);
    tree_free(T, parent);
    tree_refit_up(T, grand);
  } else {
    T.set_root(sibling); T.set_parent(sibling, -1);
    tree_free(T, parent);
  }
}
// b2DynamicTree::CreateProxy / the re-insertion half of MoveProxy for vehicle i whose fattened box is fat[4 i ..]
__device__ void tree_create_proxy(Tree T, const float* fat, int i) {
  const int id = tree_alloc(T);                        // == tree_leaf_of(i) by construction
  float* b = T.bb(id);
  b[0] = fat[4 * i]; b[1] = fat[4 * i + 1]; b[2] = fat[4 * i + 2]; b[3] = fat[4 * i + 3];
  tree_insert_leaf(T, id);
}
__device__ void tree_move_proxy(Tree T, const float* fat, int i) {
  const int id = tree_leaf_of(i);
  tree_remove_leaf(T, id);
  float* b = T.bb(id);
  b[0] = fat[4 * i]; b[1] = fat[4 * i + 1]; b[2] = fat[4 * i + 2]; b[3] = fat[4 * i + 3];
  tree_insert_leaf(T, id);
}

// b2BroadPhase::UpdatePairs + QueryCallback + b2ContactManager::AddPair.  tail: the scenario's broad-phase state (CS_TAIL layout),
// global or its LDS copy.  One tree query per buffered proxy, hits in the tree's stack order (child2 before child1); contacts are
// created in (move-buffer order, hit order) — that order is what the creation stamps record and the solver later follows.
// Serial form (one lane); stack: 64 ints of LDS (a private array would live in scratch memory: ~500 cycles per push / pop).
__device__ void find_new_contacts(float* cs, float* tail, int N, int NP, Tree T, int* stack) {
  float* fat = tail + CS_TAIL;
  float* moved = fat + 4 * N;
  float* move_buf = moved + N + 3 * N;
  const int n_move = (int)tail[3];
  int stamp = (int)tail[1];
  for (int k = 0; k < n_move; ++k) {
    const int q = (int)move_buf[k];
    const float* fq = fat + 4 * q;
    int sc = 0;
    stack[sc++] = T.root();
    while (sc > 0) {
      const int id = stack[--sc];
      if (id < 0 || !aabb_overlap(T.bb(id), fq)) continue;
      if (!T.leaf(id)) { if (sc + 2 <= 64) { stack[sc++] = T.c1(id); stack[sc++] = T.c2(id); } continue; }
      const int o = tree_veh_of(id);
      if (o == q) continue;
      if (moved[o] != 0.f && o > q) continue;
      const int i = o < q ? o : q, j = o < q ? q : o;
      float* m = cs + (size_t)(i * (2 * N - i - 1) / 2 + (j - i - 1)) * CS_STRIDE;
      if (m[CS_EXISTS] != 0.f) continue;
      for (int z = 0; z < CS_STRIDE; ++z) m[z] = 0.f;
      m[CS_EXISTS] = 1.f;
      m[CS_STAMP] = (float)(++stamp);
    }
  }
  for (int k = 0; k < n_move; ++k) moved[(int)move_buf[k]] = 0.f;
  tail[1] = (float)stamp;
  tail[3] = 0.f;
}
// The same, called by ALL lanes of one wave: the queries are read-only walks of the tree, so lane k walks it for buffered proxy
// k (its stack and hit list in `scratch`: QW_STACK ints + QW_HITS bytes per lane) and lane 0 then creates the contacts in the
// serial order.  More than 64 buffered proxies, a deeper stack or more hits than the lists hold: the serial form.
#define QW_STACK 24
#define QW_HITS 32
#define QW_BYTES (64 * QW_STACK * 4 + 64 * QW_HITS + 64 * 4)
__device__ void find_new_contacts_wave(float* cs, float* tail, int N, int NP, Tree T, void* scratch, int* serial_stack) {
  const int lane = threadIdx.x & 63;
  int* qstack = static_cast<int*>(scratch) + lane * QW_STACK;
  unsigned char* qhits = reinterpret_cast<unsigned char*>(static_cast<int*>(scratch) + 64 * QW_STACK);
  int* nhits = reinterpret_cast<int*>(qhits + 64 * QW_HITS);
  float* fat = tail + CS_TAIL;
  float* moved = fat + 4 * N;
  float* move_buf = moved + N + 3 * N;
  const int n_move = (int)tail[3];
  bool bad = n_move > 64;
  if (!bad && lane < n_move) {
    const int q = (int)move_buf[lane];
    const float* fq = fat + 4 * q;
    int sc = 0, nh = 0;
    qstack[sc++] = T.root();
    while (sc > 0) {
      const int id = qstack[--sc];
      if (id < 0 || !aabb_overlap(T.bb(id), fq)) continue;
      if (!T.leaf(id)) {
        if (sc + 2 <= QW_STACK) { qstack[sc++] = T.c1(id); qstack[sc++] = T.c2(id); } else bad = true;
        continue;
      }
      if (nh < QW_HITS) qhits[lane * QW_HITS + nh++] = (unsigned char)tree_veh_of(id); else bad = true;
    }
    nhits[lane] = nh;
  }
  bad = __any(bad);
  __builtin_amdgcn_wave_barrier();
  if (lane != 0) return;
  if (bad) { find_new_contacts(cs, tail, N, NP, T, serial_stack); return; }
  int stamp = (int)tail[1];
  for (int k = 0; k < n_move; ++k) {
    const int q = (int)move_buf[k];
    for (int h = 0; h < nhits[k]; ++h) {
      const int o = qhits[k * QW_HITS + h];
      if (o == q) continue;
      if (moved[o] != 0.f && o > q) continue;
      const int i = o < q ? o : q, j = o < q ? q : o;
      float* m = cs + (size_t)(i * (2 * N - i - 1) / 2 + (j - i - 1)) * CS_STRIDE;
      if (m[CS_EXISTS] != 0.f) continue;
      for (int z = 0; z < CS_STRIDE; ++z) m[z] = 0.f;
      m[CS_EXISTS] = 1.f;
      m[CS_STAMP] = (float)(++stamp);
    }
  }
  for (int k = 0; k < n_move; ++k) moved[(int)move_buf[k]] = 0.f;
  tail[1] = (float)stamp;
  tail[3] = 0.f;
}
__device__ __forceinline__ void buffer_move(float* tail, int N, int i) {
  float* moved = tail + CS_TAIL + 4 * N;
  float* move_buf = moved + N + 3 * N;
  const int n = (int)tail[3];
  moved[i] = 1.f;
  if (n < 4 * N) { move_buf[n] = (float)i; tail[3] = (float)(n + 1); }
}

// per-scenario body arrays in LDS, shared by the phases of one step
struct BodyLds {
  float cx[64], cy[64], a[64], vx[64], vy[64], w[64], sleep[64], lcx[64], lcy[64], invm[64], invi[64], px[64], py[64];
  int awake[64];
  unsigned long long adj[64];
};

// the touching graph: bit j of adj[i] and bit i of adj[j]
__device__ __forceinline__ void adj_link(BodyLds& B, int i, int j) {
  atomicOr(&B.adj[i], 1ull << j);
  atomicOr(&B.adj[j], 1ull << i);
}
// one island with contacts, solved by a single lane (b2Island::Solve + b2ContactSolver)
struct Constraint {
  int ia, ib, count, vcount;
  float* m;
  V2 normal, rA[2], rB[2];
  float nmass[2], tmass[2], nimp[2], timp[2], K[4], NM[4];
};
__device__ void island_solve(BodyLds& B, const int* bodies, int nb, Constraint* C, int nc, float h, float dt_ratio,
                             V2* pc, float* pa, V2* vv, float* vw) {
  for (int i = 0; i < nb; ++i) {            // gravity / forces / damping are zero: v += +0 (a -0 becomes +0), * 1.0f
    const int b = bodies[i];
    pc[i] = v2(B.cx[b], B.cy[b]); pa[i] = B.a[b]; vv[i] = v2(B.vx[b] + 0.0f, B.vy[b] + 0.0f); vw[i] = B.w[b] + 0.0f;
  }
  const float friction = sqrtf(0.2f * 0.2f);
  for (int c = 0; c < nc; ++c) {            // constructor + InitializeVelocityConstraints
    Constraint& k = C[c];
    const int gA = bodies[k.ia], gB = bodies[k.ib];
    const float mA = B.invm[gA], mB = B.invm[gB], iA = B.invi[gA], iB = B.invi[gB];
    k.count = k.vcount = (int)k.m[CS_COUNT];
    for (int j = 0; j < k.count; ++j) { k.nimp[j] = dt_ratio * k.m[5 * j + 2]; k.timp[j] = dt_ratio * k.m[5 * j + 3]; }
    const V2 cAv = pc[k.ia], cBv = pc[k.ib];
    Xf xfA, xfB;
    xfA.q.s = sinf(pa[k.ia]); xfA.q.c = cosf(pa[k.ia]);
    xfB.q.s = sinf(pa[k.ib]); xfB.q.c = cosf(pa[k.ib]);
    { const V2 r = rot_mul(xfA.q, v2(B.lcx[gA], B.lcy[gA])); xfA.p = v2(cAv.x - r.x, cAv.y - r.y); }
    { const V2 r = rot_mul(xfB.q, v2(B.lcx[gB], B.lcy[gB])); xfB.p = v2(cBv.x - r.x, cBv.y - r.y); }
    V2 wn, wp[2];
    const V2 ln = v2(k.m[CS_LN], k.m[CS_LN + 1]), lpt = v2(k.m[CS_LP], k.m[CS_LP + 1]);
    if (k.m[CS_TYPE] == 1.0f) {
      wn = rot_mul(xfA.q, ln);
      const V2 pp = xf_mul(xfA, lpt);
      for (int j = 0; j < k.count; ++j) {
        const V2 cp = xf_mul(xfB, v2(k.m[5 * j], k.m[5 * j + 1]));
        const float t = B2_POLY_RADIUS - dot2(v2(cp.x - pp.x, cp.y - pp.y), wn);
        const V2 a = v2(cp.x + t * wn.x, cp.y + t * wn.y);
        const V2 b = v2(cp.x - B2_POLY_RADIUS * wn.x, cp.y - B2_POLY_RADIUS * wn.y);
        wp[j] = v2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
      }
    } else {
      wn = rot_mul(xfB.q, ln);
      const V2 pp = xf_mul(xfB, lpt);
      for (int j = 0; j < k.count; ++j) {
        const V2 cp = xf_mul(xfA, v2(k.m[5 * j], k.m[5 * j + 1]));
        const float t = B2_POLY_RADIUS - dot2(v2(cp.x - pp.x, cp.y - pp.y), wn);
        const V2 b = v2(cp.x + t * wn.x, cp.y + t * wn.y);
        const V2 a = v2(cp.x - B2_POLY_RADIUS * wn.x, cp.y - B2_POLY_RADIUS * wn.y);
        wp[j] = v2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
      }
      wn = v2(-wn.x, -wn.y);
    }
    k.normal = wn;
    for (int j = 0; j < k.count; ++j) {
      k.rA[j] = v2(wp[j].x - cAv.x, wp[j].y - cAv.y);
      k.rB[j] = v2(wp[j].x - cBv.x, wp[j].y - cBv.y);
      const float rnA = crossvv(k.rA[j], wn), rnB = crossvv(k.rB[j], wn);
      const float kn = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
      k.nmass[j] = kn > 0.0f ? 1.0f / kn : 0.0f;
      const V2 tg = cross_vs(wn, 1.0f);
      const float rtA = crossvv(k.rA[j], tg), rtB = crossvv(k.rB[j], tg);
      const float kt = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
      k.tmass[j] = kt > 0.0f ? 1.0f / kt : 0.0f;
    }
    if (k.vcount == 2) {
      const float rn1A = crossvv(k.rA[0], wn), rn1B = crossvv(k.rB[0], wn);
      const float rn2A = crossvv(k.rA[1], wn), rn2B = crossvv(k.rB[1], wn);
      const float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
      const float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
      const float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
      if (k11 * k11 < 1000.0f * (k11 * k22 - k12 * k12)) {
        k.K[0] = k11; k.K[1] = k12; k.K[2] = k12; k.K[3] = k22;
        float det = k11 * k22 - k12 * k12;
        if (det != 0.0f) det = 1.0f / det;
        k.NM[0] = det * k22; k.NM[2] = -det * k12; k.NM[1] = -det * k12; k.NM[3] = det * k11;
      } else {
        k.vcount = 1;
      }
    }
  }
  for (int c = 0; c < nc; ++c) {            // WarmStart
    Constraint& k = C[c];
    const int gA = bodies[k.ia], gB = bodies[k.ib];
    const float mA = B.invm[gA], mB = B.invm[gB], iA = B.invi[gA], iB = B.invi[gB];
    V2 vA = vv[k.ia], vB = vv[k.ib]; float wA = vw[k.ia], wB = vw[k.ib];
    const V2 nrm = k.normal, tg = cross_vs(nrm, 1.0f);
    for (int j = 0; j < k.vcount; ++j) {
      const V2 P = v2(k.nimp[j] * nrm.x + k.timp[j] * tg.x, k.nimp[j] * nrm.y + k.timp[j] * tg.y);
      wA -= iA * crossvv(k.rA[j], P);
      vA.x -= mA * P.x; vA.y -= mA * P.y;
      wB += iB * crossvv(k.rB[j], P);
      vB.x += mB * P.x; vB.y += mB * P.y;
    }
    vv[k.ia] = vA; vw[k.ia] = wA; vv[k.ib] = vB; vw[k.ib] = wB;
  }
  for (int it = 0; it < 8; ++it)            // velocity iterations
    for (int c = 0; c < nc; ++c) {
      Constraint& k = C[c];
      const int gA = bodies[k.ia], gB = bodies[k.ib];
      const float mA = B.invm[gA], mB = B.invm[gB], iA = B.invi[gA], iB = B.invi[gB];
      V2 vA = vv[k.ia], vB = vv[k.ib]; float wA = vw[k.ia], wB = vw[k.ib];
      const V2 nrm = k.normal, tg = cross_vs(nrm, 1.0f);
      auto rel_v = [&](int j) {
        const V2 cb = cross_sv(wB, k.rB[j]), ca = cross_sv(wA, k.rA[j]);
        return v2(vB.x + cb.x - vA.x - ca.x, vB.y + cb.y - vA.y - ca.y);
      };
      auto apply = [&](V2 P, int j) {
        vA.x -= mA * P.x; vA.y -= mA * P.y; wA -= iA * crossvv(k.rA[j], P);
        vB.x += mB * P.x; vB.y += mB * P.y; wB += iB * crossvv(k.rB[j], P);
      };
      for (int j = 0; j < k.vcount; ++j) {
        const V2 dv = rel_v(j);
        const float vt = dot2(dv, tg) - 0.0f;
        float lambda = k.tmass[j] * (-vt);
        const float max_f = friction * k.nimp[j];
        const float ni = b2maxf(-max_f, b2minf(k.timp[j] + lambda, max_f));
        lambda = ni - k.timp[j];
        k.timp[j] = ni;
        apply(v2(lambda * tg.x, lambda * tg.y), j);
      }
      if (k.vcount == 1) {
        const V2 dv = rel_v(0);
        const float vn = dot2(dv, nrm);
        float lambda = -k.nmass[0] * (vn - 0.0f);
        const float ni = b2maxf(k.nimp[0] + lambda, 0.0f);
        lambda = ni - k.nimp[0];
        k.nimp[0] = ni;
        apply(v2(lambda * nrm.x, lambda * nrm.y), 0);
      } else {
        const V2 a = v2(k.nimp[0], k.nimp[1]);
        const V2 dv1 = rel_v(0), dv2 = rel_v(1);
        float vn1 = dot2(dv1, nrm), vn2 = dot2(dv2, nrm);
        V2 b = v2(vn1 - 0.0f, vn2 - 0.0f);
        { const V2 Ka = v2(k.K[0] * a.x + k.K[2] * a.y, k.K[1] * a.x + k.K[3] * a.y); b.x -= Ka.x; b.y -= Ka.y; }
        V2 x;
        bool done = false;
        auto block_apply = [&]() {
          const V2 d = v2(x.x - a.x, x.y - a.y);
          const V2 P1 = v2(d.x * nrm.x, d.x * nrm.y), P2 = v2(d.y * nrm.x, d.y * nrm.y);
          vA.x -= mA * (P1.x + P2.x); vA.y -= mA * (P1.y + P2.y);
          wA -= iA * (crossvv(k.rA[0], P1) + crossvv(k.rA[1], P2));
          vB.x += mB * (P1.x + P2.x); vB.y += mB * (P1.y + P2.y);
          wB += iB * (crossvv(k.rB[0], P1) + crossvv(k.rB[1], P2));
          k.nimp[0] = x.x; k.nimp[1] = x.y;
          done = true;
        };
        { const V2 t = v2(k.NM[0] * b.x + k.NM[2] * b.y, k.NM[1] * b.x + k.NM[3] * b.y); x = v2(-t.x, -t.y); }
        if (x.x >= 0.0f && x.y >= 0.0f) block_apply();
        if (!done) {
          x.x = -k.nmass[0] * b.x; x.y = 0.0f;
          vn2 = k.K[1] * x.x + b.y;
          if (x.x >= 0.0f && vn2 >= 0.0f) block_apply();
        }
        if (!done) {
          x.x = 0.0f; x.y = -k.nmass[1] * b.y;
          vn1 = k.K[2] * x.y + b.x;
          if (x.y >= 0.0f && vn1 >= 0.0f) block_apply();
        }
        if (!done) {
          x.x = 0.0f; x.y = 0.0f;
          vn1 = b.x; vn2 = b.y;
          if (vn1 >= 0.0f && vn2 >= 0.0f) block_apply();
        }
      }
      vv[k.ia] = vA; vw[k.ia] = wA; vv[k.ib] = vB; vw[k.ib] = wB;
    }
  for (int c = 0; c < nc; ++c)              // StoreImpulses
    for (int j = 0; j < C[c].vcount; ++j) { C[c].m[5 * j + 2] = C[c].nimp[j]; C[c].m[5 * j + 3] = C[c].timp[j]; }
  for (int i = 0; i < nb; ++i) {            // integrate positions
    V2 v = vv[i]; float w = vw[i];
    const float tx = h * v.x, ty = h * v.y;
    if (tx * tx + ty * ty > B2_MAXTRANSLATION * B2_MAXTRANSLATION) {
      const float ratio = B2_MAXTRANSLATION / sqrtf(tx * tx + ty * ty);
      v.x *= ratio; v.y *= ratio;
    }
    const float rot = h * w;
    if (rot * rot > B2_MAXROTATION * B2_MAXROTATION) {
      const float ratio = B2_MAXROTATION / fabsf(rot);
      w *= ratio;
    }
    pc[i].x += h * v.x; pc[i].y += h * v.y;
    pa[i] += h * w;
    vv[i] = v; vw[i] = w;
  }
  bool position_solved = false;
  for (int it = 0; it < 3; ++it) {          // position iterations
    float min_sep = 0.0f;
    for (int c = 0; c < nc; ++c) {
      Constraint& k = C[c];
      const int gA = bodies[k.ia], gB = bodies[k.ib];
      const float mA = B.invm[gA], mB = B.invm[gB], iA = B.invi[gA], iB = B.invi[gB];
      V2 cAv = pc[k.ia], cBv = pc[k.ib]; float aA = pa[k.ia], aB = pa[k.ib];
      const V2 ln = v2(k.m[CS_LN], k.m[CS_LN + 1]), lpt = v2(k.m[CS_LP], k.m[CS_LP + 1]);
      for (int j = 0; j < k.count; ++j) {
        Xf xfA, xfB;
        xfA.q.s = sinf(aA); xfA.q.c = cosf(aA); xfB.q.s = sinf(aB); xfB.q.c = cosf(aB);
        { const V2 r = rot_mul(xfA.q, v2(B.lcx[gA], B.lcy[gA])); xfA.p = v2(cAv.x - r.x, cAv.y - r.y); }
        { const V2 r = rot_mul(xfB.q, v2(B.lcx[gB], B.lcy[gB])); xfB.p = v2(cBv.x - r.x, cBv.y - r.y); }
        V2 nrm, point; float separation;
        if (k.m[CS_TYPE] == 1.0f) {
          nrm = rot_mul(xfA.q, ln);
          const V2 pp = xf_mul(xfA, lpt);
          const V2 cp = xf_mul(xfB, v2(k.m[5 * j], k.m[5 * j + 1]));
          separation = dot2(v2(cp.x - pp.x, cp.y - pp.y), nrm) - B2_POLY_RADIUS - B2_POLY_RADIUS;
          point = cp;
        } else {
          nrm = rot_mul(xfB.q, ln);
          const V2 pp = xf_mul(xfB, lpt);
          const V2 cp = xf_mul(xfA, v2(k.m[5 * j], k.m[5 * j + 1]));
          separation = dot2(v2(cp.x - pp.x, cp.y - pp.y), nrm) - B2_POLY_RADIUS - B2_POLY_RADIUS;
          point = cp;
          nrm = v2(-nrm.x, -nrm.y);
        }
        const V2 rA = v2(point.x - cAv.x, point.y - cAv.y), rB = v2(point.x - cBv.x, point.y - cBv.y);
        min_sep = b2minf(min_sep, separation);
        const float Cc = b2maxf(-0.2f, b2minf(0.2f * (separation + B2_LINEAR_SLOP), 0.0f));
        const float rnA = crossvv(rA, nrm), rnB = crossvv(rB, nrm);
        const float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
        const float impulse = K > 0.0f ? -Cc / K : 0.0f;
        const V2 P = v2(impulse * nrm.x, impulse * nrm.y);
        cAv.x -= mA * P.x; cAv.y -= mA * P.y; aA -= iA * crossvv(rA, P);
        cBv.x += mB * P.x; cBv.y += mB * P.y; aB += iB * crossvv(rB, P);
      }
      pc[k.ia] = cAv; pa[k.ia] = aA; pc[k.ib] = cBv; pa[k.ib] = aB;
    }
    if (min_sep >= -3.0f * B2_LINEAR_SLOP) { position_solved = true; break; }
  }
  float min_sleep = B2_FLT_MAX;              // copy back, SynchronizeTransform, sleep
  for (int i = 0; i < nb; ++i) {
    const int b = bodies[i];
    B.cx[b] = pc[i].x; B.cy[b] = pc[i].y; B.a[b] = pa[i]; B.vx[b] = vv[i].x; B.vy[b] = vv[i].y; B.w[b] = vw[i];
    const float qs = sinf(pa[i]), qc = cosf(pa[i]);
    B.px[b] = B.cx[b] - (qc * B.lcx[b] - qs * B.lcy[b]);
    B.py[b] = B.cy[b] - (qs * B.lcx[b] + qc * B.lcy[b]);
    if (B.w[b] * B.w[b] > B2_ANGSLEEPTOL * B2_ANGSLEEPTOL ||
        B.vx[b] * B.vx[b] + B.vy[b] * B.vy[b] > B2_LINSLEEPTOL * B2_LINSLEEPTOL) {
      B.sleep[b] = 0.0f; min_sleep = 0.0f;
    } else {
      B.sleep[b] += h; min_sleep = b2minf(min_sleep, B.sleep[b]);
    }
  }
  if (min_sleep >= B2_TIMETOSLEEP && position_solved)
    for (int i = 0; i < nb; ++i) {
      const int b = bodies[i];
      B.awake[b] = 0; B.sleep[b] = 0.0f; B.vx[b] = B.vy[b] = 0.0f; B.w[b] = 0.0f;
    }
}

// init_pose [S,N,4] = x, y, heading, speed; size [S,N,2] = length, width
__global__ __launch_bounds__(256) void sim_init_kernel(int N, int E, const float* __restrict__ init_pose,
                                                       const float* __restrict__ size, const float* __restrict__ edges,
                                                       const unsigned char* __restrict__ exists,
                                                       float* __restrict__ phys, float* __restrict__ hist_states,
                                                       unsigned char* __restrict__ coll, int Tmax1,
                                                       float* __restrict__ contact_state) {
  const int NP = N * (N - 1) / 2, per = CS_PER(N);
  float* cs = contact_state ? contact_state + (size_t)blockIdx.x * per : nullptr;
  if (cs) {                                              // no contacts, no impulses, b2World::m_inv_dt0 = 0
    for (int i = threadIdx.x; i < per; i += blockDim.x) cs[i] = 0.f;
  }
  __shared__ float corner[64][8];
  __shared__ __attribute__((aligned(16))) float box[64][4];
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
  if (cs && tid == 0) {
    // proxies in creation order (vehicle.cc:137-179): CreateFixture on the body at the origin (b2DynamicTree::CreateProxy),
    // SetAngle and SetPosition (b2Body::SetTransform -> Synchronize); contacts are looked for at the first step
    float* tail = cs + (size_t)NP * CS_STRIDE;
    float* fat = tail + CS_TAIL;
    Tree T{fat + 12 * N, tail + 4};
    tree_reset(T, N);
    for (int i = 0; i < N; ++i) {
      const float* p = phys + ((size_t)s * N + i) * PHYS_STRIDE;
      const Box b = box_of(size[((size_t)s * N + i) * 2 + 1], size[((size_t)s * N + i) * 2]);
      Xf xf0; xf0.p = v2(0.f, 0.f); xf0.q.s = sinf(0.f); xf0.q.c = cosf(0.f);
      float bb[4];
      shape_aabb(b, xf0, bb);
      fat[4 * i] = bb[0] - B2_AABB_EXT; fat[4 * i + 1] = bb[1] - B2_AABB_EXT;
      fat[4 * i + 2] = bb[2] + B2_AABB_EXT; fat[4 * i + 3] = bb[3] + B2_AABB_EXT;
      tree_create_proxy(T, fat, i);
      buffer_move(tail, N, i);
      Xf xfa; xfa.p = v2(0.f, 0.f); xfa.q.s = sinf(p[P_A]); xfa.q.c = cosf(p[P_A]);
      if (synchronize_fixture(fat + 4 * i, b, xfa, xfa)) { tree_move_proxy(T, fat, i); buffer_move(tail, N, i); }
      Xf xfp = xfa; xfp.p = v2(p[P_PX], p[P_PY]);
      if (synchronize_fixture(fat + 4 * i, b, xfp, xfp)) { tree_move_proxy(T, fat, i); buffer_move(tail, N, i); }
    }
    tail[2] = 1.f;                                       // m_newContacts
  }
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
                                                       int t, int Tmax1, float dt, int kinematic,
                                                       float* __restrict__ contact_state, int s_base,
                                                       int* __restrict__ guard, const float* __restrict__ expert) {
  __shared__ BodyLds B;
  __shared__ int isl_bodies[64], isl_stack[64], isl_index[64], wake[64], tele[64], in_isl[64], moved_now[64], expt[64];
  __shared__ V2 isl_pc[64], isl_vv[64];
  __shared__ float isl_pa[64], isl_vw[64];
  __shared__ Constraint isl_c[MAX_ISLAND_CONTACTS];
  static_assert(sizeof(Constraint) * MAX_ISLAND_CONTACTS >= QW_BYTES, "the pair search borrows the constraint array between island solves");
  __shared__ int isl_b0[64], isl_nb[64], isl_c0[64], isl_nc[64], isl_count;    // islands of this step: spans in isl_bodies / isl_c
  __shared__ float pair_stamp[64 * 63 / 2];                                     // creation stamp of every TOUCHING contact (LDS copy for the DFS)
  __shared__ unsigned char pair_mark[64 * 63 / 2];                              // contact already taken into an island
  __shared__ float corner[64][8];
  __shared__ __attribute__((aligned(16))) float box[64][4];
  __shared__ int flag_veh[64], flag_edge[64];
  __shared__ float px[64], py[64], hd[64], sp[64];
  const int s = blockIdx.x + s_base, tid = threadIdx.x;     // s_base: launches are cut into chunks of <= one workgroup per CU
#ifdef SIM_TIMING
  unsigned long long tl_ = __builtin_amdgcn_s_memtime();
#endif
  if (tid < N) {
    const size_t sn = (size_t)s * N + tid;
    float* p = phys + sn * PHYS_STRIDE;
    const float L = size[sn * 2 + 0];
    double accel, steer;
    tele[tid] = 0; in_isl[tid] = 0; moved_now[tid] = 0;
    expt[tid] = (expert && expert[sn * 4] == expert[sn * 4]) ? 1 : 0;      // x = NaN: not expert-controlled in this step
    if (!exists[sn]) {                                   // autoregressive_policy.py:260-263
      accel = 0.0; steer = 0.0;
      set_transform(p, -1000000.f, -1000000.f, p[P_A]);
      p[P_TELE] = 0.f;                                   // a parked vehicle drops a pending set_position request (it must not fire on revival)
      tele[tid] = 1;
    } else {
      if (p[P_TELE] != 0.f) {                            // Vehicle::set_position -> BaseCar::SetPosition -> b2Body::SetTransform
        set_transform(p, p[P_TX], p[P_TY], p[P_A]);      // (vehicle.cc:75-87, BaseCar.cpp:28-32): before this step's controls
        p[P_TELE] = 0.f;
        tele[tid] = 1;
      }
      if (act_f64) {
      accel = act_f64[sn * 2 + 0];
      steer = act_f64[sn * 2 + 1];
    } else {                                             // dataset.py:322-338 undiscretize_actions (float64)
      const int tok = act_tok[sn];
      accel = (double)(tok / dz.n_steer) / (double)(dz.n_accel - 1);
      steer = (double)(tok % dz.n_steer) / (double)(dz.n_steer - 1);
      accel = accel * (dz.max_accel - dz.min_accel) + dz.min_accel;
      steer = steer * (dz.max_steer - dz.min_steer) + dz.min_steer;
    }
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
      // body state into LDS for the Box2D step below
      B.cx[tid] = p[P_CX]; B.cy[tid] = p[P_CY]; B.a[tid] = p[P_A]; B.vx[tid] = nvx; B.vy[tid] = nvy; B.w[tid] = ang;
      B.sleep[tid] = sleep_t; B.awake[tid] = awake != 0.f; B.lcx[tid] = p[P_LCX]; B.lcy[tid] = p[P_LCY];
      B.px[tid] = p[P_PX]; B.py[tid] = p[P_PY];
      B.adj[tid] = 0ull; wake[tid] = 0;
      float lx, ly;
      local_center(size[sn * 2 + 1], L, &lx, &ly, &B.invm[tid], &B.invi[tid]);
    }
  }
  if (!kinematic) {
    // ================================================================================================ b2World::Step
    const int NP = N * (N - 1) / 2;
    float* cs = contact_state ? contact_state + (size_t)s * CS_PER(N) : nullptr;
    // the scenario's broad-phase state (CS_TAIL words, fat boxes, moved flags, sweep origins, move buffer, the dynamic tree) works
    // from LDS for the whole step — the serial parts (proxy moves, pair search) touch nothing else — and is written back at the end
    __shared__ float tail_lds[CS_TAIL + 12 * 64 + 2 * 64 * TN_STRIDE];
    float* gtail = cs ? cs + (size_t)NP * CS_STRIDE : nullptr;
    const int n_tail = CS_TAIL + 12 * N + 2 * N * TN_STRIDE;
    float* tail = tail_lds;
    float* fat = tail + CS_TAIL;
    float* sweep0 = fat + 4 * N + N;
    // the dynamic tree of this world works from LDS (one lane walks it; written back at the end of the step)
    Tree T{fat + 12 * N, tail + 4};
    if (cs)
      for (int i = tid; i < n_tail; i += blockDim.x) tail_lds[i] = gtail[i];
    SYNCJ();
    SIMT(0)
    if (cs) {
      if (tid == 0) {
        // teleported (non-existing) vehicles: b2Body::SetTransform synchronises the proxy and flags new contacts
        for (int i = 0; i < N; ++i)
          if (tele[i]) {
            const size_t si = (size_t)s * N + i;
            const Box b = box_of(size[si * 2 + 1], size[si * 2]);
            Xf xf; xf.p = v2(B.px[i], B.py[i]); xf.q.s = sinf(B.a[i]); xf.q.c = cosf(B.a[i]);
            if (synchronize_fixture(fat + 4 * i, b, xf, xf)) { tree_move_proxy(T, fat, i); buffer_move(tail, N, i); }
            tail[2] = 1.f;
          }
      }
      if (tid < 64) {                                      // wave 0 (lane 0's writes above are in program order for its own wave)
        __builtin_amdgcn_wave_barrier();
        if (tail[2] != 0.f) find_new_contacts_wave(cs, tail, N, NP, T, isl_c, isl_stack);
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) tail[2] = 0.f;
      }
      SYNCJ();
      SIMT(1)
      // ---- b2ContactManager::Collide / b2Contact::Update over the existing contacts (i < j: fixture A = i, B = j)
      for (int pr = tid; pr < NP; pr += blockDim.x) {
        float* m = cs + (size_t)pr * CS_STRIDE;
        pair_mark[pr] = 0;
        if (m[CS_EXISTS] == 0.f) continue;
        int i = 0, rem = pr;                              // pair index -> (i, j), rows of lengths N-1, N-2, ...
        while (rem >= N - 1 - i) { rem -= N - 1 - i; ++i; }
        const int j = i + 1 + rem;
        if (!B.awake[i] && !B.awake[j]) continue;
        if (!aabb_overlap(fat + 4 * i, fat + 4 * j)) {   // the fat AABBs ceased to overlap: the contact is destroyed
          for (int z = 0; z < CS_STRIDE; ++z) m[z] = 0.f;
          continue;
        }
        float old[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) old[k] = m[k];
        const int old_count = (int)m[CS_COUNT];
        const bool was_touching = m[CS_TOUCH] != 0.f;
        Xf xfA, xfB;
        xfA.p = v2(B.px[i], B.py[i]); xfA.q.s = sinf(B.a[i]); xfA.q.c = cosf(B.a[i]);
        xfB.p = v2(B.px[j], B.py[j]); xfB.q.s = sinf(B.a[j]); xfB.q.c = cosf(B.a[j]);
        const size_t si = (size_t)s * N + i, sj = (size_t)s * N + j;
        const Box bA = box_of(size[si * 2 + 1], size[si * 2]), bB = box_of(size[sj * 2 + 1], size[sj * 2]);
        const int cnt = collide_boxes(m, bA, xfA, bB, xfB);
        for (int k = 0; k < cnt; ++k) {                    // match old contact ids, copy the stored impulses
          float ni = 0.f, ti = 0.f;
          const unsigned id2 = __float_as_uint(m[5 * k + 4]);
          for (int l = 0; l < old_count; ++l)
            if (__float_as_uint(old[5 * l + 4]) == id2) { ni = old[5 * l + 2]; ti = old[5 * l + 3]; break; }
          m[5 * k + 2] = ni; m[5 * k + 3] = ti;
        }
        m[CS_COUNT] = (float)cnt;
        if (cnt == 0) { m[CS_COUNT] = 0.f; }
        const bool touching = cnt > 0;
        m[CS_TOUCH] = touching ? 1.f : 0.f;
        if (touching != was_touching) { wake[i] = 1; wake[j] = 1; }
        if (touching) { adj_link(B, i, j); pair_stamp[pr] = m[CS_STAMP]; }
      }
      SYNCJ();
      // pairs of two sleeping bodies were skipped above: their (unchanged) touching flag still links them
      for (int pr = tid; pr < NP; pr += blockDim.x) {
        int i = 0, rem = pr;
        while (rem >= N - 1 - i) { rem -= N - 1 - i; ++i; }
        const int j = i + 1 + rem;
        if (!B.awake[i] && !B.awake[j] && cs[(size_t)pr * CS_STRIDE + CS_TOUCH] != 0.f) {
          adj_link(B, i, j);
          pair_stamp[pr] = cs[(size_t)pr * CS_STRIDE + CS_STAMP];
        }
      }
      // the loop above must see the awake flags the FIRST loop saw (it links exactly the pairs that one skipped): the wake-ups
      // are applied behind a barrier — without it a fast wave 0 could wake bodies while other waves still read the flags
      SYNCJ();
      if (tid < N && wake[tid]) { B.awake[tid] = 1; B.sleep[tid] = 0.f; }     // b2Body::SetAwake(true)
      if (tid < N) { sweep0[3 * tid] = B.cx[tid]; sweep0[3 * tid + 1] = B.cy[tid]; sweep0[3 * tid + 2] = B.a[tid]; }
      SYNCJ();
      SIMT(2)
    }
    // ---- islands of one body: integrate on their own lanes (b2Island::Solve without contacts)
    if (tid < N && B.adj[tid] == 0ull && B.awake[tid]) {
      in_isl[tid] = 1;
      float vx = B.vx[tid] + 0.0f, vy = B.vy[tid] + 0.0f, w = B.w[tid] + 0.0f;
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
      B.cx[tid] += dt * vx;
      B.cy[tid] += dt * vy;
      B.a[tid] += dt * w;
      float sleep_t = B.sleep[tid];
      if (w * w > B2_ANGSLEEPTOL * B2_ANGSLEEPTOL || vx * vx + vy * vy > B2_LINSLEEPTOL * B2_LINSLEEPTOL) sleep_t = 0.0f;
      else sleep_t += dt;
      const float qs = sinf(B.a[tid]), qc = cosf(B.a[tid]);                    // SynchronizeTransform
      B.px[tid] = B.cx[tid] - (qc * B.lcx[tid] - qs * B.lcy[tid]);
      B.py[tid] = B.cy[tid] - (qs * B.lcx[tid] + qc * B.lcy[tid]);
      if (sleep_t >= B2_TIMETOSLEEP) { B.awake[tid] = 0; sleep_t = 0.f; vx = vy = 0.f; w = 0.f; }
      B.vx[tid] = vx; B.vy[tid] = vy; B.w[tid] = w; B.sleep[tid] = sleep_t;
    }
    SIMT(3)
    // ---- islands with contacts.  Lane 0 runs b2World::Solve's depth-first search (island membership and, inside an island, the
    // order of bodies and contacts are Box2D's: newest body first, newest contact first) over LDS copies of the contact stamps;
    // the islands themselves are independent — disjoint bodies, disjoint contacts — so each is then solved by its own lane of
    // wave 0 with the sequential-impulse solver, all of them in lockstep (the time of the largest island, not the sum).
    if (cs && tid == 0) {
      unsigned long long in_island = 0ull;
      int n_isl = 0, boff = 0, coff = 0;
      for (int seed = N - 1; seed >= 0; --seed) {          // m_bodyList: newest body first
        if (B.adj[seed] == 0ull || ((in_island >> seed) & 1ull) || !B.awake[seed]) continue;
        int nb = 0, nc = 0, sc = 0;
        isl_stack[sc++] = seed; in_island |= 1ull << seed;
        while (sc > 0) {
          const int b = isl_stack[--sc];
          isl_index[b] = nb; isl_bodies[boff + nb++] = b;
          B.awake[b] = 1;                                  // woken without resetting the sleep timer
          in_isl[b] = 1;
          unsigned long long rem_edges = B.adj[b];         // touching contacts of b, newest contact first
          while (rem_edges) {
            int o = -1; float best = -1.f;
            for (unsigned long long r2 = rem_edges; r2; r2 &= r2 - 1) {
              const int c = __ffsll((long long)r2) - 1;
              const int i2 = b < c ? b : c, j2 = b < c ? c : b;
              const float st = pair_stamp[i2 * (2 * N - i2 - 1) / 2 + (j2 - i2 - 1)];
              if (st > best) { best = st; o = c; }
            }
            rem_edges &= ~(1ull << o);
            const int i = b < o ? b : o, j = b < o ? o : b;
            const int pr = i * (2 * N - i - 1) / 2 + (j - i - 1);   // rows i of length N-1-i
            if (pair_mark[pr]) continue;                   // already in this island
            pair_mark[pr] = 1;
            if (coff + nc < MAX_ISLAND_CONTACTS) {         // (the touching graph of disjoint boxes is planar: <= 3 N - 6 contacts)
              isl_c[coff + nc].m = cs + (size_t)pr * CS_STRIDE; isl_c[coff + nc].ia = i; isl_c[coff + nc].ib = j; ++nc;
            } else if (guard) {
              atomicAdd(guard, 1);                         // deeply overlapping boxes: the contact is NOT solved — counted, never silent
                                                           // (guard = the SIMULATOR word of the guard pair, common.h)
            }
            if ((in_island >> o) & 1ull) continue;
            isl_stack[sc++] = o; in_island |= 1ull << o;
          }
        }
        for (int c = 0; c < nc; ++c) {
          isl_c[coff + c].ia = isl_index[isl_c[coff + c].ia]; isl_c[coff + c].ib = isl_index[isl_c[coff + c].ib];
        }
        isl_b0[n_isl] = boff; isl_nb[n_isl] = nb; isl_c0[n_isl] = coff; isl_nc[n_isl] = nc;
        ++n_isl; boff += nb; coff += nc;
      }
      isl_count = n_isl;
    }
    SYNCJ();
    if (cs && tid < isl_count) {
      const float dt_ratio = tail[0] * dt;
      const int b0 = isl_b0[tid], c0 = isl_c0[tid];
      island_solve(B, isl_bodies + b0, isl_nb[tid], isl_c + c0, isl_nc[tid], dt, dt_ratio, isl_pc + b0, isl_pa + b0, isl_vv + b0,
                   isl_vw + b0);
    }
    SYNCJ();
    if (cs && tid == 0) tail[0] = dt > 0.0f ? 1.0f / dt : 0.0f;                                   // m_inv_dt0
    SIMT(4)
    SYNCJ();
    if (cs) {
      // ---- b2Body::SynchronizeFixtures of every body that was in an island, then b2ContactManager::FindNewContacts
      if (tid < N && in_isl[tid]) {
        const size_t si = (size_t)s * N + tid;
        const Box b = box_of(size[si * 2 + 1], size[si * 2]);
        Xf xf2; xf2.p = v2(B.px[tid], B.py[tid]); xf2.q.s = sinf(B.a[tid]); xf2.q.c = cosf(B.a[tid]);
        Xf xf1 = xf2;
        if (B.awake[tid]) {
          xf1.q.s = sinf(sweep0[3 * tid + 2]); xf1.q.c = cosf(sweep0[3 * tid + 2]);
          const V2 r = rot_mul(xf1.q, v2(B.lcx[tid], B.lcy[tid]));
          xf1.p = v2(sweep0[3 * tid] - r.x, sweep0[3 * tid + 1] - r.y);
        }
        moved_now[tid] = synchronize_fixture(fat + 4 * tid, b, xf1, xf2) ? 1 : 0;
      }
      SYNCJ();
      SIMT(5)
      if (tid == 0) {
        for (int b = N - 1; b >= 0; --b)                   // m_bodyList order: newest body first
          if (moved_now[b]) { tree_move_proxy(T, fat, b); buffer_move(tail, N, b); }
        SIMT(6)
      }
      if (tid < 64) {
        __builtin_amdgcn_wave_barrier();
        find_new_contacts_wave(cs, tail, N, NP, T, isl_c, isl_stack);
        SIMT(7)
      }
      SYNCJ();
    }
    // ---- Scenario::Step for EXPERT-CONTROLLED objects (nocturne/cpp/src/scenario.cc:276-283), after the world step, in object order:
    // Vehicle::set_position (vehicle.cc:82-87: b2Body::SetTransform at the current angle), set_heading (vehicle.cc:89-94: SetTransform
    // at the current position, angle = heading - pi/2), set_speed (vehicle.cc:96-105: SetLinearVelocity, which wakes the body when the
    // velocity is not zero).  Each SetTransform synchronises the proxy (the dynamic tree's shape decides the order of later contacts)
    // and flags the new-contact search of the next step's top.
    if (expert && tid == 0) {
      for (int i = 0; i < N; ++i) {
        if (!expt[i]) continue;
        const size_t si = (size_t)s * N + i;
        const float ex = expert[si * 4], ey = expert[si * 4 + 1], eh = expert[si * 4 + 2], es = expert[si * 4 + 3];
        const Box b = box_of(size[si * 2 + 1], size[si * 2]);
        for (int pass = 0; pass < 2; ++pass) {
          const float ang = pass == 0 ? B.a[i] : (float)((double)eh - M_PI_D * 0.5f);
          const float qs = sinf(ang), qc = cosf(ang);
          B.px[i] = ex; B.py[i] = ey; B.a[i] = ang;
          B.cx[i] = (qc * B.lcx[i] - qs * B.lcy[i]) + ex;
          B.cy[i] = (qs * B.lcx[i] + qc * B.lcy[i]) + ey;
          if (cs) {
            Xf xf; xf.p = v2(ex, ey); xf.q.s = qs; xf.q.c = qc;
            if (synchronize_fixture(fat + 4 * i, b, xf, xf)) { tree_move_proxy(T, fat, i); buffer_move(tail, N, i); }
            tail[2] = 1.f;                                           // b2World::m_newContacts
          }
        }
        const float nvx = es * cosf(eh), nvy = es * sinf(eh);
        if (nvx * nvx + nvy * nvy > 0.0f) { B.awake[i] = 1; B.sleep[i] = 0.f; }
        B.vx[i] = nvx; B.vy[i] = nvy;
      }
    }
    SYNCJ();
    if (cs)
      for (int i = tid; i < n_tail; i += blockDim.x) gtail[i] = tail_lds[i];
    if (tid < N) {
      const size_t sn = (size_t)s * N + tid;
      float* p = phys + sn * PHYS_STRIDE;
      p[P_CX] = B.cx[tid]; p[P_CY] = B.cy[tid]; p[P_A] = B.a[tid]; p[P_PX] = B.px[tid]; p[P_PY] = B.py[tid];
      p[P_VX] = B.vx[tid]; p[P_VY] = B.vy[tid]; p[P_W] = B.w[tid];
      p[P_AWAKE] = B.awake[tid] ? 1.f : 0.f; p[P_SLEEP] = B.sleep[tid];
      // ---- Vehicle::Step read-back (vehicle.cc:45-55)
      px[tid] = B.px[tid];
      py[tid] = B.py[tid];
      sp[tid] = sqrtf(B.vx[tid] * B.vx[tid] + B.vy[tid] * B.vy[tid]);
      hd[tid] = (float)((double)B.a[tid] + M_PI_D * 0.5f);
      if (expt[tid]) {                                    // Object::heading_ / speed_ of an expert-controlled object are the logged values
        const size_t se = ((size_t)s * N + tid) * 4;
        hd[tid] = expert[se + 2];
        sp[tid] = expert[se + 3];
      }
      p[P_HEADING] = hd[tid]; p[P_SPEED] = sp[tid];
    }
  }
  SYNCJ();
  SIMT(8)
  collide_and_record(s, N, E, size, edges, exists, hist_states, coll, t + 1, Tmax1, corner, box, flag_veh, flag_edge, px,
                     py, hd, sp);
  SIMT(9)
}

int launch_sim_init(int S, int N, int E, const float* init_pose, const float* size, const float* edges,
                    const unsigned char* exists, float* phys, float* hist_states, unsigned char* coll, int Tmax1,
                    float* contact_state, hipStream_t st) {
  if (S <= 0) return CTRLSIM_OK;
  if (N < 1 || N > 64 || E < 0) return CTRLSIM_EINVAL;
  hipLaunchKernelGGL(sim_init_kernel, dim3(S), dim3(256), 0, st, N, E, init_pose, size, edges, exists, phys, hist_states,
                     coll, Tmax1, contact_state);
  return ctrlsim_launch_status();
}

// Vehicle::set_position(x, y) for the vehicles whose xy entry is not NaN: the request is parked in the body record and applied
// (b2Body::SetTransform, proxy synchronisation, new-contact search) at the top of the next step, before that step's controls
__global__ void sim_set_position_kernel(int n, const float* __restrict__ xy, float* __restrict__ phys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = xy[2 * i], y = xy[2 * i + 1];
  if (x != x || y != y) return;
  float* p = phys + (size_t)i * PHYS_STRIDE;
  p[P_TX] = x; p[P_TY] = y; p[P_TELE] = 1.f;
}
int launch_sim_set_position(int S, int N, const float* xy, float* phys, hipStream_t st) {
  if (S <= 0) return CTRLSIM_OK;
  if (N < 1 || N > 64 || !xy || !phys) return CTRLSIM_EINVAL;
  hipLaunchKernelGGL(sim_set_position_kernel, dim3((S * N + 255) / 256), dim3(256), 0, st, S * N, xy, phys);
  return ctrlsim_launch_status();
}

int launch_sim_step(int S, int N, int E, const int* act_tok, const double* act_f64, const double* disc6,
                    const float* size, const float* edges, const unsigned char* exists, float* phys,
                    float* hist_states, unsigned char* coll, double* applied, int t, int Tmax1, float dt, int kinematic,
                    float* contact_state, const float* expert, hipStream_t st) {
  if (S <= 0) return CTRLSIM_OK;
  if (N < 1 || N > 64 || E < 0 || t < 0 || t + 1 >= Tmax1 || (!act_tok && !act_f64)) return CTRLSIM_EINVAL;
  if (expert && kinematic) return CTRLSIM_EINVAL;        // expert replay is defined on the FreeCar / Box2D integrator (scenario.cc:266-292)
  SimDiscretisation dz{disc6[0], disc6[1], disc6[2], disc6[3], (int)disc6[4], (int)disc6[5]};
  prof_before(PROF_SIM, st);
  // The step takes a CU's whole LDS (its own ~60 KB + a dynamic remainder it never touches): a scenario's workgroup then never
  // shares a CU with LDS-using workgroups of kernels running on other streams.  Sharing one with the split-operand matrix kernels made rollouts non-reproducible (5 of 40
  // provoked runs differ without this, 0 of 40 with it); the step is latency-bound with one workgroup per scenario, so the
  // exclusivity costs nothing.
  static const int n_cus = [] {
    int dev = 0, n = 0;
    return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ? n : 0;
  }();
  static const int excl_lds = [] {                   // gfx950: 160 KB of LDS per CU, all of it available to one workgroup
    hipFuncAttributes fa;
    if (getenv("CTRLSIM_SIM_SHARED_CU")) return 0;   // A/B switch: the step shares CUs like any other kernel
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&sim_step_kernel)) != hipSuccess) return 0;
    const long dyn = (160L * 1024 - (long)fa.sharedSizeBytes - 2048) & ~255L;
    if (dyn <= 0 || hipFuncSetAttribute(reinterpret_cast<const void*>(&sim_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)dyn) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    return (int)dyn;
  }();
  // More scenarios than CUs: the launch is cut into chunks of n_cus workgroups (back to back on the stream), so that the
  // exclusivity holds whatever the batch size — a single 1024-workgroup launch would have to share CUs (four per CU fit).
  const int chunk = (excl_lds > 0 && n_cus > 0) ? n_cus : S;
  for (int s0 = 0; s0 < S; s0 += chunk) {
    const int n = S - s0 < chunk ? S - s0 : chunk;
    hipLaunchKernelGGL(sim_step_kernel, dim3(n), dim3(256), excl_lds, st, N, E, act_tok, act_f64, dz, size, edges, exists, phys,
                       hist_states, coll, applied, t, Tmax1, dt, kinematic, contact_state, s0, ctrlsim_simguard_ptr(), expert);
  }
  // per scenario: body + control state in and out (20 floats), one history row + flags out, the road-edge segments in,
  // and (contacts) the persistent Box2D state in and out (20 floats per vehicle pair + broad phase)
  prof_after(PROF_SIM, 0.0, st, (double)S * (N * (2.0 * 80 + 32 + 2 + 4) + 16.0 * E +
                                             (contact_state ? 8.0 * (N * (N - 1) / 2 * 20 + 6 + 28 * N) : 0.0)));
  return ctrlsim_launch_status();
}
