// Focal grouping and agent-local context construction for S scenarios (gfx950).  -ffp-contract=off.
//
// Reference (host NumPy, float64) -> what these kernels replace:
//   policies/autoregressive_policy.py:51-165  get_data(): context window [0,T) / [t-T+1,t], greedy focal grouping with the
//                                             list-mutated-while-iterated quirk (:124-127), membership persistence
//   datasets/rl_waymo/dataset.py:278-319      select_relevant_agents: A nearest within 60 m at t=0 (slots in ascending
//                                             global index), later the persisted set minus agents that left the disc
//   datasets/rl_waymo/dataset.py:390-428      normalize_scene: SE(2) frame of the focal agent at window index 0,
//                                             nearest-P polyline selection (argsort of the max existing-point distance)
//   utils/geometry.py:14-19,30-47             angle_sub_tensor, apply_se2_transform
//   datasets/rl_waymo/dataset.py:361-387 + autoregressive_policy.py:73-78 are folded away: the engine keeps action
//   tokens / RTG bins as integers (round trips are identities, asserted in tests/test_oracle_pinned.py).
// Vehicle sets are 64-bit masks (N <= 64 vehicles per scenario = one wavefront; ballot gives the set, popcount the slot).
// All geometry is float64 exactly as in the reference, results are rounded once to float32 (`.float()` in
// modules/encoder.py:85,91-92,114 and modules/map_encoder.py:35-36).
#include "common.h"

#define TWO_PI_D 6.283185307179586476925286766559
#define PI_D 3.14159265358979323846

__device__ __forceinline__ double py_mod_2pi(double a) {   // numpy/python float %: result takes the divisor's sign
  double m = fmod(a, TWO_PI_D);
  if (m != 0.0) { if (m < 0.0) m += TWO_PI_D; } else { m = 0.0; }
  return m;
}
__device__ __forceinline__ double angle_sub(double current, double target) {   // utils/geometry.py:14-19
  double d = py_mod_2pi(target - current);
  if (d > PI_D) d = -(TWO_PI_D - d);
  return d;
}

// ------------------------------------------------------------------------------------------------ grouping
// One wavefront per scenario.  eval_order [S,N]: vehicles_to_evaluate in processing order, -1 padded.
// persist [S,N] u64 in/out.  Outputs per scenario: n_groups, grp_focal/ids/members [S,N], own_g/mem_g [S,N], tilted [S,N].
__global__ __launch_bounds__(64) void group_build_kernel(int N, int A, int T, int t, int Tmax1, double dist_thresh,
                                                         const float* __restrict__ hist_states,
                                                         const int* __restrict__ eval_order, int has_roads,
                                                         unsigned long long* __restrict__ persist,
                                                         int* __restrict__ n_groups, int* __restrict__ grp_focal,
                                                         unsigned long long* __restrict__ grp_ids,
                                                         unsigned long long* __restrict__ grp_members,
                                                         int* __restrict__ own_g, int* __restrict__ mem_g,
                                                         unsigned char* __restrict__ tilted) {
  __shared__ int todo[64];
  const int s = blockIdx.x, lane = threadIdx.x;
  const int w0 = t < T ? 0 : t - (T - 1);
  const bool live = lane < N;
  // window-index-0 position and current existence of "my" vehicle
  double mx = 0.0, my = 0.0;
  bool exist_now = false;
  if (live) {
    const float* r0 = hist_states + (((size_t)s * N + lane) * Tmax1 + w0) * 8;
    mx = (double)r0[0];
    my = (double)r0[1];
    exist_now = hist_states[(((size_t)s * N + lane) * Tmax1 + t) * 8 + 7] != 0.f;
  }
  const unsigned long long exist_mask = __ballot(exist_now);
  unsigned long long my_persist = live ? persist[(size_t)s * N + lane] : 0ull;
  int len = 0;
  {
    const int e = live ? eval_order[(size_t)s * N + lane] : -1;
    todo[lane] = e;
    len = __popcll(__ballot(e >= 0));      // entries are packed at the front
  }
  int my_own = -1, my_mem = -1;
  unsigned char my_tilt = 0;
  int g = 0;
  __syncthreads();
  while (len > 0) {
    const int focal = todo[0];
    __syncthreads();
    {  // pop front
      const int nxt = (lane + 1 < 64) ? todo[(lane + 1) & 63] : -1;
      __syncthreads();
      todo[lane] = (lane + 1 < len) ? nxt : -1;
      --len;
    }
    __syncthreads();
    if (!((exist_mask >> focal) & 1ull) || !has_roads) continue;   // dead_agent_veh_ids: zero action (mem_g stays -1)
    // ---- select_relevant_agents
    const double fx = __shfl(mx, focal, 64), fy = __shfl(my, focal, 64);
    const double dx = fx - mx, dy = fy - my;
    const double d = live ? sqrt(dx * dx + dy * dy) : __builtin_inf();
    const unsigned long long valid = __ballot(live && d < dist_thresh);
    unsigned long long ids;
    const unsigned long long fp = __shfl(my_persist, focal, 64);
    if (t == 0 || fp == 0ull) {
      int rank = 0;                                   // position in np.argsort(dist)
      for (int i = 0; i < N; ++i) {
        const double di = __shfl(d, i, 64);
        rank += (di < d || (di == d && i < lane)) ? 1 : 0;
      }
      ids = __ballot(live && rank < A) & valid;
    } else {
      ids = fp & valid;
    }
    // ---- fold still-unaccounted evaluated vehicles of this context into the group (list mutated while iterated)
    unsigned long long members = 1ull << focal;
    int i = 0;
    while (i < len) {
      const int v = todo[i];
      __syncthreads();
      if ((ids >> v) & 1ull) {
        members |= 1ull << v;
        const int nxt = todo[(lane + 1) & 63];
        __syncthreads();
        if (lane >= i) todo[lane] = (lane + 1 < len) ? nxt : -1;
        --len;
        __syncthreads();
      }
      ++i;
    }
    // ---- persist membership, RTG ownership (first group in order that contains the vehicle), action membership
    if ((members >> lane) & 1ull) { my_persist = ids; my_mem = g; }
    if (((ids >> lane) & 1ull) && my_own < 0) { my_own = g; my_tilt = (unsigned char)((members >> lane) & 1ull); }
    if (lane == 0) {
      grp_focal[(size_t)s * N + g] = focal;
      grp_ids[(size_t)s * N + g] = ids;
      grp_members[(size_t)s * N + g] = members;
    }
    ++g;
  }
  if (live) {
    persist[(size_t)s * N + lane] = my_persist;
    own_g[(size_t)s * N + lane] = my_own;
    mem_g[(size_t)s * N + lane] = my_mem;
    tilted[(size_t)s * N + lane] = my_tilt;
  }
  if (lane == 0) n_groups[s] = g;
}

// Did the focal groups of scenarios [0, S) change against a snapshot (ref_*)?  flag[0] |= 1 if any scenario's group count, a
// focal vehicle or a context membership mask differs.  The K/V-cached phase (engine.py) keeps a chunk on the incremental
// forward only while its context set is the one the cache was built for; the flag travels to the host with the group
// counts in one small asynchronous copy instead of a blocking comparison.
__global__ __launch_bounds__(256) void groups_changed_kernel(int S, int N, const int* __restrict__ n_groups,
                                                             const int* __restrict__ grp_focal,
                                                             const unsigned long long* __restrict__ grp_ids,
                                                             const int* __restrict__ ref_n, const int* __restrict__ ref_focal,
                                                             const unsigned long long* __restrict__ ref_ids,
                                                             int* __restrict__ flag) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= S * N) return;
  const int s = k / N, g = k - s * N;
  const int n = n_groups[s];
  bool diff = (g == 0 && n != ref_n[s]);
  if (g < n && (grp_focal[k] != ref_focal[k] || grp_ids[k] != ref_ids[k])) diff = true;
  if (diff) atomicOr(flag, 1);
}

// Flat context list for scenarios [s0, s1): ctx_base by exclusive scan (one block), then per-vehicle owner/member
// context ids and slots.  ctx index is local to the chunk (0-based at s0).
__global__ __launch_bounds__(256) void ctx_index_kernel(int s0, int s1, int N, const int* __restrict__ n_groups,
                                                        const int* __restrict__ grp_focal,
                                                        const unsigned long long* __restrict__ grp_ids,
                                                        const int* __restrict__ own_g, const int* __restrict__ mem_g,
                                                        int* __restrict__ ctx_scn, int* __restrict__ ctx_grp,
                                                        int* __restrict__ own_ctx, int* __restrict__ own_slot,
                                                        int* __restrict__ mem_ctx, int* __restrict__ mem_slot,
                                                        int* __restrict__ ctx_base_out) {
  __shared__ int base[4096];
  const int ns = s1 - s0;
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int i = 0; i < ns; ++i) { base[i] = acc; acc += n_groups[s0 + i]; }
    base[ns] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ns; i += blockDim.x) {
    ctx_base_out[s0 + i] = base[i];
    for (int g = 0; g < n_groups[s0 + i]; ++g) { ctx_scn[base[i] + g] = s0 + i; ctx_grp[base[i] + g] = g; }
  }
  for (int k = threadIdx.x; k < ns * N; k += blockDim.x) {
    const int i = k / N, v = k - i * N;
    const size_t sv = (size_t)(s0 + i) * N + v;
    const int og = own_g[sv], mg = mem_g[sv];
    const unsigned long long below = (v == 0) ? 0ull : (~0ull >> (64 - v));
    own_ctx[sv] = og < 0 ? -1 : base[i] + og;
    own_slot[sv] = og < 0 ? -1 : __popcll(grp_ids[(size_t)(s0 + i) * N + og] & below);
    mem_ctx[sv] = mg < 0 ? -1 : base[i] + mg;
    mem_slot[sv] = mg < 0 ? -1 : __popcll(grp_ids[(size_t)(s0 + i) * N + mg] & below);
  }
}

// Compact contexts (forward.hip: Shape): a context with n vehicles is evaluated with the smallest slot count of `sizes`
// (ascending, nb <= MAXC, last = A) that is >= n + 1 — or A itself.  hist[s, k] = focal groups of scenario s that fall into
// size class k: what the host needs (with n_groups) to cut a step into model batches.
__device__ __forceinline__ int size_class(int n, const int* sizes, int nb) {
  int k = 0;
  while (k < nb - 1 && sizes[k] < n + 1) ++k;
  return k;
}
struct SizeClasses { int nb; int sizes[MAXC]; };
__global__ __launch_bounds__(256) void group_size_hist_kernel(int S, int N, const int* __restrict__ n_groups,
                                                              const unsigned long long* __restrict__ grp_ids, SizeClasses sc,
                                                              int* __restrict__ hist) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  int h[MAXC] = {};
  for (int g = 0; g < n_groups[s]; ++g) ++h[size_class(__popcll(grp_ids[(size_t)s * N + g]), sc.sizes, sc.nb)];
  for (int k = 0; k < sc.nb; ++k) hist[(size_t)s * sc.nb + k] = h[k];
}

// ctx_index with the contexts of scenarios [s0, s1) SORTED by size class (stable in (scenario, group) order inside a class):
// class k occupies contexts [start_k, start_k + count_k).  ctx_row0[c] = first logits row of context c when every class
// writes (slots - 1, or A for the last class) rows per context, classes in order.  One block of 256 threads: thread i owns
// scenarios i, i + 256, ... of the chunk in turn; per round a block-wide exclusive scan of the threads' per-class counts gives
// every scenario its first context of each class (<= 4095 scenarios per chunk: 16 rounds).
__global__ __launch_bounds__(256) void ctx_index_classes_kernel(int s0, int s1, int N, int A, const int* __restrict__ n_groups,
                                                                const unsigned long long* __restrict__ grp_ids,
                                                                const int* __restrict__ own_g, const int* __restrict__ mem_g,
                                                                SizeClasses sc, int* __restrict__ ctx_scn,
                                                                int* __restrict__ ctx_grp, int* __restrict__ ctx_row0,
                                                                int* __restrict__ ctx_of_group,   // [S, N] scratch
                                                                int* __restrict__ own_ctx, int* __restrict__ own_slot,
                                                                int* __restrict__ mem_ctx, int* __restrict__ mem_slot) {
  __shared__ int scan[MAXC][256];
  __shared__ int total[MAXC], start[MAXC], row0[MAXC], carry[MAXC];
  const int ns = s1 - s0, tid = threadIdx.x, nb = sc.nb;
  // ---- class totals of the chunk
  int cnt[MAXC] = {};
  for (int i = tid; i < ns; i += 256)
    for (int g = 0; g < n_groups[s0 + i]; ++g) ++cnt[size_class(__popcll(grp_ids[(size_t)(s0 + i) * N + g]), sc.sizes, nb)];
  if (tid < MAXC) total[tid] = 0;
  __syncthreads();
  for (int k = 0; k < nb; ++k)
    if (cnt[k]) atomicAdd(&total[k], cnt[k]);
  __syncthreads();
  if (tid == 0) {
    int acc = 0, racc = 0;
    for (int k = 0; k < nb; ++k) {
      start[k] = acc; row0[k] = racc; carry[k] = 0;
      acc += total[k];
      racc += total[k] * (sc.sizes[k] < A ? sc.sizes[k] - 1 : A);
    }
  }
  __syncthreads();
  // ---- rounds of 256 scenarios, in scenario order
  for (int base = 0; base < ns; base += 256) {
    const int i = base + tid;
    int mine[MAXC] = {};
    if (i < ns)
      for (int g = 0; g < n_groups[s0 + i]; ++g) ++mine[size_class(__popcll(grp_ids[(size_t)(s0 + i) * N + g]), sc.sizes, nb)];
    // inclusive scan over the 256 threads per class: inside a wave by lane shuffles (6 steps, no barrier), then the three
    // preceding waves' totals through LDS — one barrier per round instead of sixteen (the 16-class kernel had grown to 0.2 ms)
    const int lane = tid & 63, wv = tid >> 6;
    int incl[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      int v = k < nb ? mine[k] : 0;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(v, off, 64);
        if (lane >= off) v += u;
      }
      incl[k] = v;
      if (lane == 63) scan[k][wv] = v;                  // the wave's total
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      int before = 0;
      for (int q = 0; q < wv; ++q) before += scan[k][q];
      incl[k] += before;
    }
    if (tid == 255)
      for (int k = 0; k < nb; ++k) scan[k][255] = incl[k];  // block total of the round (read below as scan[k][255])
    if (i < ns) {
      int fill[MAXC];
#pragma unroll
      for (int k = 0; k < MAXC; ++k) fill[k] = k < nb ? carry[k] + incl[k] - mine[k] : 0;
      for (int g = 0; g < n_groups[s0 + i]; ++g) {
        const int k = size_class(__popcll(grp_ids[(size_t)(s0 + i) * N + g]), sc.sizes, nb);
        const int c = start[k] + fill[k];
        ctx_scn[c] = s0 + i; ctx_grp[c] = g;
        ctx_row0[c] = row0[k] + fill[k] * (sc.sizes[k] < A ? sc.sizes[k] - 1 : A);
        ctx_of_group[(size_t)(s0 + i) * N + g] = c;
        ++fill[k];
      }
    }
    __syncthreads();
    if (tid < nb) carry[tid] += scan[tid][255];
    __syncthreads();
  }
  for (int k = tid; k < ns * N; k += blockDim.x) {
    const int i = k / N, v = k - i * N;
    const size_t sv = (size_t)(s0 + i) * N + v;
    const int og = own_g[sv], mg = mem_g[sv];
    const unsigned long long below = (v == 0) ? 0ull : (~0ull >> (64 - v));
    own_ctx[sv] = og < 0 ? -1 : ctx_of_group[(size_t)(s0 + i) * N + og];
    own_slot[sv] = og < 0 ? -1 : __popcll(grp_ids[(size_t)(s0 + i) * N + og] & below);
    mem_ctx[sv] = mg < 0 ? -1 : ctx_of_group[(size_t)(s0 + i) * N + mg];
    mem_slot[sv] = mg < 0 ? -1 : __popcll(grp_ids[(size_t)(s0 + i) * N + mg] & below);
  }
}

// ------------------------------------------------------------------------------------------------ context tensors
struct CtxOut {
  float* st12;            // [B, Tq, A, 12]  x,y,vx,vy,yaw,len,wid + 5 type one-hot (-1 padded slots)
  float* exist;           // [B, Tq, A]
  float* goal5;           // [B, A, 5]
  int* act_tok;           // [B, Tq, A]
  int* rtg_bin;           // [B, Tq, A, 3]
  int* tstep;             // [B, Tq]
  int* slot_gid;          // [B, A]   global vehicle index per slot, -1 = padded
  float* road_pts;        // [B, P, NP, 3]
  float* road_types;      // [B, P, 8]
};

// Up to MAXC classes of contexts (different slot counts A, separate output arrays) in ONE launch: class k holds the contexts
// [c0[k], c0[k+1]) of the batch's context list.  One launch per class left most of the chip idle — a class of a model batch is
// 50-150 contexts = workgroups, and a workgroup's float64 chain takes ~150 us whatever the grid.
struct CtxBatch { int n; int c0[MAXC + 1]; int A[MAXC]; CtxOut o[MAXC]; };

// One block per context.  Agent part: threads over (tt, slot).  Road part: two sweeps over P_all x NP points.
// Window rows [tt_first, Tq) are emitted (Tn = Tq - tt_first rows per context): the cached incremental forward only needs
// the last one or two timesteps; tt_first = 0 gives the whole window.
__global__ __launch_bounds__(256) void build_context_kernel(
    int N, int T, int t, int Tq, int tt_first, int Tmax1, int Tmax, int P_all, int P, int NP,
    const int* __restrict__ ctx_scn, const int* __restrict__ ctx_grp, const int* __restrict__ grp_focal,
    const unsigned long long* __restrict__ grp_ids, const float* __restrict__ hist_states,
    const int* __restrict__ hist_tok, const int* __restrict__ hist_rtg, const double* __restrict__ goals,  // [S,N,5] f64
    const float* __restrict__ types,                                                                     // [S,N,5]
    const float* __restrict__ roads, const float* __restrict__ rtypes,                                    // [S,P_all,NP,3], [S,P_all,8]
    int zero_tok, int zr0, int zr1, int zr2, CtxBatch cb) {
  extern __shared__ double far_[];                 // [P_all] distance key, then int rank/sel arrays behind it
  __shared__ int slot_of[64];
  __shared__ int gid_of[64];
  const int tid = threadIdx.x;
  int k_ = 0;
  while (k_ + 1 < cb.n && (int)blockIdx.x >= cb.c0[k_ + 1]) ++k_;      // wave-uniform: the class of this context
  const int A = cb.A[k_], b = blockIdx.x - cb.c0[k_];                  // b: index within the class's output arrays
  const CtxOut o = cb.o[k_];
  const int s = ctx_scn[blockIdx.x], g = ctx_grp[blockIdx.x];
  const int focal = grp_focal[(size_t)s * N + g];
  const unsigned long long ids = grp_ids[(size_t)s * N + g];
  const int n_ids = __popcll(ids);
  const int w0 = t < T ? 0 : t - (T - 1);
  if (tid < 64) {
    const unsigned long long below = (tid == 0) ? 0ull : (~0ull >> (64 - tid));
    slot_of[tid] = ((ids >> tid) & 1ull) ? __popcll(ids & below) : -1;
  }
  __syncthreads();
  if (tid < 64 && tid < N && slot_of[tid] >= 0) gid_of[slot_of[tid]] = tid;
  __syncthreads();
  if (tid < A) o.slot_gid[(size_t)b * A + tid] = tid < n_ids ? gid_of[tid] : -1;

  // frame of the focal agent at window index 0 (dataset.py:392-396)
  const float* f0 = hist_states + (((size_t)s * N + focal) * Tmax1 + w0) * 8;
  const double yaw0 = (double)f0[4];
  const double sgn = (-yaw0 > 0.0) ? 1.0 : ((-yaw0 < 0.0) ? -1.0 : 0.0);
  const double rot = (PI_D / 2) + sgn * fabs(yaw0);
  const double cr = cos(rot), sr = sin(rot);
  const double tx = (double)f0[0], ty = (double)f0[1];

  // ---- agents
  const int Tn = Tq - tt_first;
  for (int k = tid; k < Tn * A; k += blockDim.x) {
    const int to = k / A, slot = k - to * A, tt = tt_first + to;
    double raw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float ty5[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};
    // padded slots: zero action -> placeholder token, zero (un-normalised) rtg rows -> bins (0,0,0) (dataset.py:284-288)
    int tok = zero_tok, r0 = 0, r1 = 0, r2 = 0;
    if (slot < n_ids) {
      r0 = zr0; r1 = zr1; r2 = zr2;
      const int v = gid_of[slot];
      const int abs_t = w0 + tt;
      const float* row = hist_states + (((size_t)s * N + v) * Tmax1 + abs_t) * 8;
#pragma unroll
      for (int c = 0; c < 8; ++c) raw[c] = (double)row[c];
#pragma unroll
      for (int c = 0; c < 5; ++c) ty5[c] = types[((size_t)s * N + v) * 5 + c];
      if (abs_t < Tmax) {
        tok = hist_tok[((size_t)s * N + v) * Tmax + abs_t];
        const int* rb = hist_rtg + (((size_t)s * N + v) * Tmax + abs_t) * 3;
        r0 = rb[0]; r1 = rb[1]; r2 = rb[2];
      }
    }
    const double px = raw[0] - tx, py = raw[1] - ty;
    float* so = o.st12 + (((size_t)b * Tn + to) * A + slot) * 12;
    so[0] = (float)(cr * px + (-sr) * py);
    so[1] = (float)(sr * px + cr * py);
    so[2] = (float)(cr * raw[2] + (-sr) * raw[3]);
    so[3] = (float)(sr * raw[2] + cr * raw[3]);
    so[4] = (float)angle_sub(raw[4], -rot);
    so[5] = (float)raw[5];
    so[6] = (float)raw[6];
#pragma unroll
    for (int c = 0; c < 5; ++c) so[7 + c] = ty5[c];
    o.exist[((size_t)b * Tn + to) * A + slot] = (float)raw[7];
    o.act_tok[((size_t)b * Tn + to) * A + slot] = tok;
    int* rbo = o.rtg_bin + (((size_t)b * Tn + to) * A + slot) * 3;
    rbo[0] = r0; rbo[1] = r1; rbo[2] = r2;
  }
  // timesteps: self.timesteps[0, window] — rows not yet written are 0 (policy.py:53,79)
  for (int to = tid; to < Tn; to += blockDim.x) {
    const int abs_t = w0 + tt_first + to;
    o.tstep[(size_t)b * Tn + to] = abs_t <= t ? abs_t : 0;
  }
  // goals (goal row at window index 0; constant in time): dataset.py:408-415
  for (int slot = tid; slot < A; slot += blockDim.x) {
    double gr[5] = {0, 0, 0, 0, 0};
    if (slot < n_ids) {
      const double* gp = goals + ((size_t)s * N + gid_of[slot]) * 5;
#pragma unroll
      for (int c = 0; c < 5; ++c) gr[c] = gp[c];
    }
    const double px = gr[0] - tx, py = gr[1] - ty;
    float* go = o.goal5 + ((size_t)b * A + slot) * 5;
    go[0] = (float)(cr * px + (-sr) * py);
    go[1] = (float)(sr * px + cr * py);
    go[2] = (float)(cr * gr[2] + (-sr) * gr[3]);
    go[3] = (float)(sr * gr[2] + cr * gr[3]);
    go[4] = (float)angle_sub(gr[4], -rot);
  }

  // ---- roads
  const float* rsrc = roads + (size_t)s * P_all * NP * 3;
  int* sel = reinterpret_cast<int*>(far_ + P_all);     // sel[r] = source polyline of output row r
  if (P_all > P) {
    for (int p = tid; p < P_all; p += blockDim.x) {
      double mxd = 0.0;
      bool first = true;
      for (int q = 0; q < NP; ++q) {
        const float* pt = rsrc + ((size_t)p * NP + q) * 3;
        const double px = (double)pt[0] - tx, py = (double)pt[1] - ty;
        const double x = cr * px + (-sr) * py, y = sr * px + cr * py;
        const double dd = sqrt(x * x + y * y) * (double)pt[2];
        if (first || dd > mxd) { mxd = dd; first = false; }
      }
      far_[p] = mxd;
    }
    __syncthreads();
    for (int p = tid; p < P_all; p += blockDim.x) {
      const double d = far_[p];
      int rank = 0;
      for (int q = 0; q < P_all; ++q) {
        const double dq = far_[q];
        rank += (dq < d || (dq == d && q < p)) ? 1 : 0;
      }
      if (rank < P) sel[rank] = p;
    }
    __syncthreads();
  }
  const int n_live = P_all > P ? P : P_all;
  for (int k = tid; k < P * NP; k += blockDim.x) {
    const int r = k / NP, q = k - r * NP;
    float* po = o.road_pts + (((size_t)b * P + r) * NP + q) * 3;
    if (r < n_live) {
      const int p = P_all > P ? sel[r] : r;
      const float* pt = rsrc + ((size_t)p * NP + q) * 3;
      const double px = (double)pt[0] - tx, py = (double)pt[1] - ty;
      po[0] = (float)(cr * px + (-sr) * py);
      po[1] = (float)(sr * px + cr * py);
      po[2] = pt[2];
    } else {
      po[0] = 0.f; po[1] = 0.f; po[2] = 0.f;
    }
  }
  for (int k = tid; k < P * 8; k += blockDim.x) {
    const int r = k >> 3, c = k & 7;
    float v = -1.f;
    if (r < n_live) {
      const int p = P_all > P ? sel[r] : r;
      v = rtypes[((size_t)s * P_all + p) * 8 + c];
    }
    o.road_types[((size_t)b * P + r) * 8 + c] = v;
  }
}

// ------------------------------------------------------------------------------------------------ launchers
int launch_group_build(int S, int N, int A, int T, int t, int Tmax1, double dist_thresh, const float* hist_states,
                       const int* eval_order, int has_roads, unsigned long long* persist, int* n_groups, int* grp_focal,
                       unsigned long long* grp_ids, unsigned long long* grp_members, int* own_g, int* mem_g,
                       unsigned char* tilted, hipStream_t st) {
  if (S <= 0) return CTRLSIM_OK;
  if (N < 1 || N > 64 || A < 1 || A > 64) return CTRLSIM_EINVAL;
  hipLaunchKernelGGL(group_build_kernel, dim3(S), dim3(64), 0, st, N, A, T, t, Tmax1, dist_thresh, hist_states, eval_order,
                     has_roads, persist, n_groups, grp_focal, grp_ids, grp_members, own_g, mem_g, tilted);
  return ctrlsim_launch_status();
}

int launch_ctx_index(int s0, int s1, int N, const int* n_groups, const int* grp_focal, const unsigned long long* grp_ids,
                     const int* own_g, const int* mem_g, int* ctx_scn, int* ctx_grp, int* own_ctx, int* own_slot,
                     int* mem_ctx, int* mem_slot, int* ctx_base, hipStream_t st) {
  if (s1 <= s0) return CTRLSIM_OK;
  if (s1 - s0 > 4095) return CTRLSIM_EINVAL;
  hipLaunchKernelGGL(ctx_index_kernel, dim3(1), dim3(256), 0, st, s0, s1, N, n_groups, grp_focal, grp_ids, own_g, mem_g,
                     ctx_scn, ctx_grp, own_ctx, own_slot, mem_ctx, mem_slot, ctx_base);
  return ctrlsim_launch_status();
}

int launch_build_context_classes(int n, const int* Bk, const int* Ak, const CtxOut* ok, int N, int T, int t, int Tq, int tt_first,
                                 int Tmax1, int Tmax, int P_all, int P, int NP, const int* ctx_scn, const int* ctx_grp,
                                 const int* grp_focal, const unsigned long long* grp_ids, const float* hist_states,
                                 const int* hist_tok, const int* hist_rtg, const double* goals, const float* types,
                                 const float* roads, const float* rtypes, const int* zero4, hipStream_t st) {
  if (n < 1 || n > MAXC || N > 64 || Tq < 1 || tt_first < 0 || tt_first >= Tq) return CTRLSIM_EINVAL;
  CtxBatch cb;
  cb.n = 0; cb.c0[0] = 0;
  double bytes = 0.0;
  for (int k = 0; k < n; ++k) {
    if (Bk[k] < 0 || Ak[k] < 1 || Ak[k] > 64) return CTRLSIM_EINVAL;
    if (Bk[k] == 0) continue;
    cb.A[cb.n] = Ak[k]; cb.o[cb.n] = ok[k];
    cb.c0[cb.n + 1] = cb.c0[cb.n] + Bk[k];
    ++cb.n;
    // per context: the scenario's road points in once (P_all x NP x 12 B; the selection sweep re-reads them from cache), the P
    // selected polylines + types out, the window rows of A agents in (8 floats + token + 3 bins) and out (12 floats + 5 ints)
    bytes += (double)Bk[k] * (12.0 * P_all * NP + 12.0 * P * NP + 32.0 * P + (double)(Tq - tt_first) * Ak[k] * (48.0 + 68.0));
  }
  if (cb.n == 0) return CTRLSIM_OK;
  const size_t shm = (size_t)P_all * sizeof(double) + (size_t)(P > 0 ? P : 1) * sizeof(int);
  prof_before(PROF_CTX, st);
  hipLaunchKernelGGL(build_context_kernel, dim3(cb.c0[cb.n]), dim3(256), shm, st, N, T, t, Tq, tt_first, Tmax1, Tmax, P_all, P, NP,
                     ctx_scn, ctx_grp, grp_focal, grp_ids, hist_states, hist_tok, hist_rtg, goals, types, roads, rtypes, zero4[0],
                     zero4[1], zero4[2], zero4[3], cb);
  prof_after(PROF_CTX, 0.0, st, bytes);
  return ctrlsim_launch_status();
}
int launch_build_context(int B, int N, int A, int T, int t, int Tq, int tt_first, int Tmax1, int Tmax, int P_all, int P, int NP,
                         const int* ctx_scn, const int* ctx_grp, const int* grp_focal,
                         const unsigned long long* grp_ids, const float* hist_states, const int* hist_tok,
                         const int* hist_rtg, const double* goals, const float* types, const float* roads,
                         const float* rtypes, const int* zero4, CtxOut o, hipStream_t st) {
  return launch_build_context_classes(1, &B, &A, &o, N, T, t, Tq, tt_first, Tmax1, Tmax, P_all, P, NP, ctx_scn, ctx_grp, grp_focal,
                                      grp_ids, hist_states, hist_tok, hist_rtg, goals, types, roads, rtypes, zero4, st);
}

int launch_groups_changed(int S, int N, const int* n_groups, const int* grp_focal, const unsigned long long* grp_ids,
                          const int* ref_n, const int* ref_focal, const unsigned long long* ref_ids, int* flag, hipStream_t st) {
  if (S <= 0) return CTRLSIM_OK;
  if (!n_groups || !grp_focal || !grp_ids || !ref_n || !ref_focal || !ref_ids || !flag || N < 1 || N > 64) return CTRLSIM_EINVAL;
  hipLaunchKernelGGL(groups_changed_kernel, dim3((S * N + 255) / 256), dim3(256), 0, st, S, N, n_groups, grp_focal, grp_ids,
                     ref_n, ref_focal, ref_ids, flag);
  return ctrlsim_launch_status();
}

int launch_group_size_hist(int S, int N, const int* n_groups, const unsigned long long* grp_ids, int nb, const int* sizes,
                           int* hist, hipStream_t st) {
  if (S <= 0) return CTRLSIM_OK;
  if (!n_groups || !grp_ids || !sizes || !hist || nb < 1 || nb > MAXC) return CTRLSIM_EINVAL;
  SizeClasses sc;
  sc.nb = nb;
  for (int k = 0; k < MAXC; ++k) sc.sizes[k] = k < nb ? sizes[k] : 0;
  hipLaunchKernelGGL(group_size_hist_kernel, dim3((S + 255) / 256), dim3(256), 0, st, S, N, n_groups, grp_ids, sc, hist);
  return ctrlsim_launch_status();
}
int launch_ctx_index_classes(int s0, int s1, int N, int A, const int* n_groups, const unsigned long long* grp_ids,
                             const int* own_g, const int* mem_g, int nb, const int* sizes, int* ctx_scn, int* ctx_grp,
                             int* ctx_row0, int* ctx_of_group, int* own_ctx, int* own_slot, int* mem_ctx, int* mem_slot,
                             hipStream_t st) {
  if (s1 <= s0) return CTRLSIM_OK;
  if (nb < 1 || nb > MAXC || !sizes || sizes[nb - 1] != A) return CTRLSIM_EINVAL;
  SizeClasses sc;
  sc.nb = nb;
  for (int k = 0; k < MAXC; ++k) sc.sizes[k] = k < nb ? sizes[k] : 0;
  hipLaunchKernelGGL(ctx_index_classes_kernel, dim3(1), dim3(256), 0, st, s0, s1, N, A, n_groups, grp_ids, own_g, mem_g, sc,
                     ctx_scn, ctx_grp, ctx_row0, ctx_of_group, own_ctx, own_slot, mem_ctx, mem_slot);
  return ctrlsim_launch_status();
}
