// Polyline (road segment) encoder front end: point MLP + single-seed multi-head attention pooling (gfx950).
//
// Reference: modules/map_encoder.py:28-53.  Each polyline's NP points (x, y, exist) go through
// road_pts_encoder = Linear(3,256)-LN-ReLU-Linear(256,256), then an 8-head attention whose ONLY query is the learned
// `map_seeds` vector pools the points (key padding = non-existing points; a polyline with no existing point un-masks
// point 0 so the softmax is defined, :31), followed by out_proj.  Because the query is a constant and everything after
// the ReLU is linear up to the softmax, the per-point work collapses (weights folded in float64 at pack time,
// ctrlsim_amd/pack.py):
//   score[pt,h] = h1[pt] . U[:,h] + c[h]            U = W2^T Wk_h^T q_h / sqrt(32),  q = Wq seed + bq
//   pooled[h]   = sum_pt softmax_pt(score[:,h]) h1[pt]
//   attn[h*32+j]= (Wv_h W2)[j,:] . pooled[h] + (Wv_h b2 + bv_h)[j]
// where h1 = ReLU(LN(W1 (x,y,e) + b1)).  The reference's two 20 000-row GEMMs per context (point MLP layer 2: 2.6 GFLOP,
// K/V projection: 5.2 GFLOP) become ~0.3 GFLOP of VALU work; results are equal in exact arithmetic.
// The first layer and its LayerNorm are evaluated in closed form too: centred, gain-scaled weights Wc and the 4x4 Gram matrix G
// of the centred weights give  LN(.)_c = Wc[c].(x,y,e,1) * rsqrt((x,y,e,1)^T G (x,y,e,1) + eps) + beta_c  with one pass over c.
// One workgroup per polyline: phase 1 thread-per-point (LN statistics + 8 scores, weights are wave-uniform scalar
// loads), phase 2 thread-per-channel (softmax-weighted pooling), phase 3 thread-per-output (256x256 folded matrix,
// stored transposed so the wave reads it coalesced from L2).
#include "common.h"
#include "classes.h"
#include <type_traits>

#define MAXNP 256
#ifndef MAP_PK
#define MAP_PK 1          // 1: the packed-fp32 kernel (round 6), 0: the scalar kernel (A/B)
#endif

struct MapPoolWeights {
  const float* Wc2;     // [128,4,2] the same with the channels of a pair interleaved per component (packed kernel: scalar register pairs)
  const float* Wc;      // [256,4]  g_c * (W1[c,:] - column mean, b1[c] - mean(b1)): LN(W1 p + b1)_c = Wc[c] . (x,y,e,1) * rstd + ln_b[c]
  const float* G;       // [10]     upper triangle of sum_c wt_c wt_c^T / 256 (wt = Wc without the gain): var = (x,y,e,1)^T G (x,y,e,1)
  const float* ln_b;    // [256]
  const float* U;       // [256,8]
  const float* cb;      // [8]
  const float* Mt;      // [256(c),256(j)]
  const float* mb;      // [256]
  int force_pad;        // 1: cfg.model.use_map = False (ctrlsim_dims.flags bit 1) — every polyline row is key-padded: the scene encoder and the
                        // decoder's memory then hold the vehicles' initial-state rows only, as modules/encoder.py:168-170 builds them
};

// G polylines per workgroup (2 when a polyline has <= 128 points).  Only VISIBLE points are evaluated: a padded point has softmax
// weight exp(-inf) = 0 and adds an exact zero to every sum, so the points of the workgroup's polylines are first compacted (order
// kept) and the thread-per-point phase, the softmax and the pooling loop run over the compact list — bit-identical results, and a
// polyline with 40 of 100 points costs 40 % of a full one (the thread-per-point phase is 60 % of the kernel's instructions).
// Classes of contexts in one launch (forward.hip: a model batch): the polylines of all classes are one flat list; only the
// padding byte goes to a class-dependent place (scene rows per context M differ with the slot count): polylines
// [bp0[k], bp0[k+1]) belong to class k, whose padding rows start at pad0[k] with M[k] rows per context.
struct MapClasses { int n; int bp0[MAXC + 1]; int M[MAXC]; long pad0[MAXC]; };

__global__ __launch_bounds__(256) void map_pool_kernel(int NP, int P, MapClasses mc, int G, int total, const float* __restrict__ road_pts,
                                                       MapPoolWeights w, float* __restrict__ attn_pre,
                                                       unsigned char* __restrict__ src_pad) {
  // the pooled vectors reuse the point / score arrays (16 KB per workgroup instead of 28: twice the resident waves to hide the
  // scalar-load latency of the weight stream); a barrier separates the last read of `sc` from the first write of `pooled`
  __shared__ float lds_[2 * 8 * DM];
  float (*pts)[3] = reinterpret_cast<float (*)[3]>(lds_);                         // visible points, compacted  [MAXNP][3]
  float* stat = lds_ + MAXNP * 3;                                                 // [MAXNP]
  float (*sc)[8] = reinterpret_cast<float (*)[8]>(lds_ + MAXNP * 4);              // [MAXNP][8]
  float (*pooled)[8][DM] = reinterpret_cast<float (*)[8][DM]>(lds_);              // [2][8][DM]
  static_assert(MAXNP * 12 <= 2 * 8 * DM, "alias layout");
  __shared__ int any_exist[2], wcnt[4], q0[3];     // q0[g] .. q0[g+1]: compact range of polyline g
  const int bp0 = blockIdx.x * G, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g_here = min(G, total - bp0);                              // polylines of this block
  const int n_pts = g_here * NP;
  const float* src = road_pts + (size_t)bp0 * NP * 3;
  if (tid < 2) any_exist[tid] = 0;
  __syncthreads();
  float x = 0.f, y = 0.f, e = 0.f;
  if (tid < n_pts) {
    x = src[tid * 3]; y = src[tid * 3 + 1]; e = src[tid * 3 + 2];
    if (e != 0.f) any_exist[tid >= NP] = 1;
  }
  __syncthreads();
  // key padding: non-existing points; a polyline without any existing point un-masks its point 0 (map_encoder.py:31)
  const int g_of = tid >= NP, p_in = tid - g_of * NP;
  const bool vis = tid < n_pts && (e != 0.f || (any_exist[g_of] == 0 && p_in == 0));
  const unsigned long long bal = __ballot(vis);
  if (lane == 0) wcnt[wv] = __popcll(bal);
  __syncthreads();
  int before = __popcll(bal & ((1ull << lane) - 1ull));
  for (int k = 0; k < wv; ++k) before += wcnt[k];
  if (vis) { pts[before][0] = x; pts[before][1] = y; pts[before][2] = e; }
  if (tid == 0) q0[0] = 0;
  if (tid == NP && g_here > 1) q0[1] = before;
  if (tid == 255) {
    const int all = before + (vis ? 1 : 0);       // lane 255 sees every earlier flag
    q0[g_here] = all;
  }
  __syncthreads();
  const int n_vis = q0[g_here];
  // ---- phase 1: per visible point LN statistics and head scores.  Round 6: the thread-per-point phase used to have work for
  // ceil(n_vis / 64) of the four waves (two on the bench's scenes: the other two sat at the barrier through the kernel's longest phase).
  // Now all four work: with nph = ceil(n_vis / 64) <= 2 point blocks the 256 channels are cut into parts = 4 / nph ranges, wave w takes
  // point block w % nph and channel range w / nph, and the partial scores meet in the softmax below (scp[part][point][head] fills the
  // same 8 KB as sc[point][head]).  Three or four point blocks: one wave each, all channels, as before.
  const int nph = (n_vis + 63) >> 6, parts = nph <= 2 ? (nph == 1 ? 4 : 2) : 1;      // workgroup-uniform
  const int wv_s = __builtin_amdgcn_readfirstlane(wv);      // scalar: the channel range must stay wave-uniform (weights by s_load)
  const int my_ph = parts > 1 ? wv_s % nph : wv_s, my_part = parts > 1 ? wv_s / nph : 0;
  const int c_lo = my_part * (DM / parts);
  const int pt1 = my_ph * 64 + lane;
  float (*scp)[8] = sc + my_part * (64 * nph);                                       // this wave's partial-score block
  if (pt1 < n_vis && my_ph < nph) {
    const int tid = pt1;                              // (the point index; shadows the thread index in this block on purpose)
    const float x = pts[tid][0], y = pts[tid][1], e = pts[tid][2];
    // LayerNorm statistics in closed form (the layer is affine in the point: pack.py); the mean drops out of the centred weights
    const float* G = w.G;
    const float var = x * (G[0] * x + 2.f * (G[1] * y + G[2] * e + G[3])) + y * (G[4] * y + 2.f * (G[5] * e + G[6])) +
                      e * (G[7] * e + 2.f * G[8]) + G[9];
    const float rstd = 1.0f / sqrtf(fmaxf(var, 0.f) + 1e-5f);
    float s8[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) s8[h] = my_part == 0 ? w.cb[h] : 0.f;
    // (compile-time trip counts: with run-time bounds hipcc gave up the unrolling / scalar-load batching of this loop and the kernel
    // lost 5-7 % — even in the unchanged one-range case)
    auto channels = [&](auto NC) {
      const float* Wc = w.Wc + c_lo * 4;
      const float* lb = w.ln_b + c_lo;
      const float* U = w.U + c_lo * 8;
      for (int c = 0; c < decltype(NC)::value; ++c) {
        const float d = fmaf(Wc[c * 4 + 2], e, fmaf(Wc[c * 4 + 1], y, fmaf(Wc[c * 4], x, Wc[c * 4 + 3])));
        const float hv = fmaxf(fmaf(d, rstd, lb[c]), 0.f);
#pragma unroll
        for (int h = 0; h < 8; ++h) s8[h] = fmaf(hv, U[c * 8 + h], s8[h]);
      }
    };
    if (parts == 1) channels(std::integral_constant<int, DM>{});
    else if (parts == 2) channels(std::integral_constant<int, DM / 2>{});
    else channels(std::integral_constant<int, DM / 4>{});
    if (my_part == 0) stat[tid] = rstd;
#pragma unroll
    for (int h = 0; h < 8; ++h) scp[tid][h] = s8[h];
  }
  __syncthreads();
  // ---- softmax over the visible points of a polyline, per head.  One 16-lane DPP row per (polyline, head) pair — 16 pairs = the 256
  // threads — each lane takes every 16th point; maximum and sum are row reductions (round 5: the loop ran on 16 LANES of one wave, three
  // dependent passes over ~60 points with an LDS round trip each, while the other three waves sat at the barrier)
  {
    const int pair = tid >> 4, sub = tid & 15, g = pair >> 3, hd = pair & 7;
    const int a = g < g_here ? q0[g] : 0, b = g < g_here ? q0[g + 1] : 0;
    float mx = -__builtin_inff();
    for (int p = a + sub; p < b; p += 16) {
      float t = sc[p][hd];
      if (parts > 1) {                                 // the channel ranges' partial scores (every (point, head) belongs to one lane)
        for (int q = 1; q < parts; ++q) t += sc[q * 64 * nph + p][hd];
        sc[p][hd] = t;
      }
      mx = fmaxf(mx, t);
    }
    mx = fmaxf(mx, dpp_f32<0xB1>(mx));       // quad_perm [1,0,3,2]
    mx = fmaxf(mx, dpp_f32<0x4E>(mx));       // quad_perm [2,3,0,1]
    mx = fmaxf(mx, dpp_f32<0x141>(mx));      // row_half_mirror
    mx = fmaxf(mx, dpp_f32<0x140>(mx));      // row_mirror: every lane of the row holds the row maximum
    float z = 0.f;
    for (int p = a + sub; p < b; p += 16) {
      const float ev = expf(sc[p][hd] - mx);
      sc[p][hd] = ev;
      z += ev;
    }
    z += dpp_f32<0xB1>(z);
    z += dpp_f32<0x4E>(z);
    z += dpp_f32<0x141>(z);
    z += dpp_f32<0x140>(z);
    const float inv = 1.0f / z;
    for (int p = a + sub; p < b; p += 16) sc[p][hd] *= inv;
  }
  __syncthreads();
  // ---- phase 2: thread = channel; pooled[g][h][c] = sum_pt a[pt,h] * h1[pt,c]
  {
    const int c = tid;
    const float w0 = w.Wc[c * 4], w1 = w.Wc[c * 4 + 1], w2 = w.Wc[c * 4 + 2], bb = w.Wc[c * 4 + 3], be = w.ln_b[c];
    float acc[2][8];
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int h = 0; h < 8; ++h) acc[g][h] = 0.f;
#ifdef MPV_ABL_NO_P2
      if (g < g_here && w0 > 1e30f)
#else
      if (g < g_here)
#endif
        for (int p = q0[g]; p < q0[g + 1]; ++p) {
          const float d = fmaf(w2, pts[p][2], fmaf(w1, pts[p][1], fmaf(w0, pts[p][0], bb)));
          const float hv = fmaxf(fmaf(d, stat[p], be), 0.f);
#pragma unroll
          for (int h = 0; h < 8; ++h) acc[g][h] = fmaf(sc[p][h], hv, acc[g][h]);
        }
    }
    __syncthreads();                       // every read of pts / stat / sc is done: `pooled` may overwrite them
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int h = 0; h < 8; ++h) pooled[g][h][c] = acc[g][h];
  }
  __syncthreads();
  // ---- phase 3: thread = output channel j of head j>>5
  {
    const int j = tid, h = j >> 5;
    // both polylines of the workgroup per pass over the folded matrix (its 256 KB come from L2 once, not once per polyline)
    float o0 = w.mb[j], o1 = o0;
#ifdef MPV_ABL_NO_P3
    for (int c = 0; c < (o0 > 1e30f ? DM : 0); ++c) {
#else
#pragma unroll 8
    for (int c = 0; c < DM; ++c) {
#endif
      const float m = w.Mt[c * DM + j];
      o0 = fmaf(pooled[0][h][c], m, o0);
      o1 = fmaf(pooled[1][h][c], m, o1);
    }
    attn_pre[(size_t)bp0 * DM + j] = o0;
    if (g_here > 1) attn_pre[(size_t)(bp0 + 1) * DM + j] = o1;
  }
  if (tid < g_here) {
    const int bp = bp0 + tid;
    int k = 0;
    while (k + 1 < mc.n && bp >= mc.bp0[k + 1]) ++k;
    const int rel = bp - mc.bp0[k], b = rel / P, p = rel - b * P;
    src_pad[mc.pad0[k] + (size_t)b * mc.M[k] + p] = (any_exist[tid] && !w.force_pad) ? 0 : 1;
  }
}


// ---- Round 6: the same function with PACKED fp32 arithmetic (v_pk_fma_f32: two fp32 FMAs per lane and instruction at the rate of one).
// The kernel above is bound by its vector instruction count (~27 VALU instructions per (point, channel): the hidden value of the point MLP
// in both the thread-per-point and the thread-per-channel phase, and eight head FMAs in each); here every FMA is one half of a packed one:
//   phase 1 (thread = point): TWO channels per trip — their four weight pairs, bias pair and score-weight pairs are scalar register pairs
//            (fold.map.Wc2: the closed-form first layer with the channels of a pair interleaved), the heads accumulate as four pairs;
//   phase 2 (thread = channel): TWO points per trip for the hidden value (the compact point list is kept as x / y / e / rstd arrays, so a
//            pair of neighbouring points is one 8-byte LDS broadcast), the eight head accumulators as four pairs per polyline;
//   phase 3 (thread = output channel): the workgroup's two polylines are the two halves (pooled[h][c][g]).
// 14 packed + 2 scalar instead of 27 scalar instructions per (point, channel); every FMA keeps its operands and its place in its
// accumulation chain, so the results are BIT-IDENTICAL to the kernel above (tests/test_gpu_ops.py compares the two).
// Only forms the co-residency finding of round 4 cleared are used: plain packed FMAs and the op_sel_hi broadcast of
// a low half; nothing reads a HIGH half into a LOW result (build.py::isa_guard refuses the object otherwise).  A polyline's compact range
// starts at an even index (8-byte LDS reads); an odd range ends in a pad point with softmax weight 0 and hidden value 0 (exact zeros).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

#define PKNP (MAXNP + 4)
__global__ __launch_bounds__(256) void map_pool_pk_kernel(int NP, int P, MapClasses mc, int G, int total, const float* __restrict__ road_pts,
                                                          MapPoolWeights w, float* __restrict__ attn_pre,
                                                          unsigned char* __restrict__ src_pad) {
  __shared__ __attribute__((aligned(16))) float lds_[2 * 8 * DM];
  float* px = lds_;                                                               // visible points, compacted: x / y / e / rstd [PKNP] each
  float* py = lds_ + PKNP;
  float* pe = lds_ + 2 * PKNP;
  float* stat = lds_ + 3 * PKNP;
  float (*sc)[8] = reinterpret_cast<float (*)[8]>(lds_ + 4 * PKNP);               // [PKNP][8]
  float (*pooled)[DM][2] = reinterpret_cast<float (*)[DM][2]>(lds_);              // [8][DM][2 polylines]  (aliases the arrays above)
  static_assert(PKNP * 12 <= 2 * 8 * DM && (PKNP & 3) == 0, "alias layout / 16-byte rows of sc");
  __shared__ int any_exist[2], wcnt[4], q0[4];     // polyline g: compact range [q0[2g], q0[2g+1])
  const int bp0 = blockIdx.x * G, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g_here = min(G, total - bp0);
  const int n_pts = g_here * NP;
  const float* src = road_pts + (size_t)bp0 * NP * 3;
  if (tid < 2) any_exist[tid] = 0;
  __syncthreads();
  float x = 0.f, y = 0.f, e = 0.f;
  if (tid < n_pts) {
    x = src[tid * 3]; y = src[tid * 3 + 1]; e = src[tid * 3 + 2];
    if (e != 0.f) any_exist[tid >= NP] = 1;
  }
  __syncthreads();
  const int g_of = tid >= NP, p_in = tid - g_of * NP;
  const bool vis = tid < n_pts && (e != 0.f || (any_exist[g_of] == 0 && p_in == 0));
  const unsigned long long bal = __ballot(vis);
  const unsigned long long bal0 = __ballot(vis && !g_of);            // polyline 0's points of this wave
  if (lane == 0) { wcnt[wv] = __popcll(bal0); }
  __shared__ int wcnt1[4];
  if (lane == 0) wcnt1[wv] = __popcll(bal) - __popcll(bal0);
  __syncthreads();
  const int n0 = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3], n1 = wcnt1[0] + wcnt1[1] + wcnt1[2] + wcnt1[3];
  const int base1 = (n0 + 1) & ~1;                                   // polyline 1 starts at an even index
  {
    const unsigned long long mine = g_of ? (bal & ~bal0) : bal0;
    int before = __popcll(mine & ((1ull << lane) - 1ull));
    for (int k = 0; k < wv; ++k) before += g_of ? wcnt1[k] : wcnt[k];
    const int slot = (g_of ? base1 : 0) + before;
    if (vis) { px[slot] = x; py[slot] = y; pe[slot] = e; }
  }
  const int end1 = base1 + n1, n_slots = (end1 + 1) & ~1;            // slots [0, n_slots): points and at most two pads
  // pads: a point with zero coordinates (finite hidden value); its softmax weights are set to zero below
  if (tid == 0 && (n0 & 1)) { px[n0] = 0.f; py[n0] = 0.f; pe[n0] = 0.f; }
  if (tid == 1 && (end1 & 1)) { px[end1] = 0.f; py[end1] = 0.f; pe[end1] = 0.f; }
  __syncthreads();
  // ---- phase 1: per point LN statistics and head scores; point blocks x channel ranges over the four waves as in the kernel above
  const int nph = (n_slots + 63) >> 6, parts = nph <= 2 ? (nph == 1 ? 4 : 2) : 1;
  const int wv_s = __builtin_amdgcn_readfirstlane(wv);
  const int my_ph = parts > 1 ? wv_s % nph : wv_s, my_part = parts > 1 ? wv_s / nph : 0;
  const int c_lo = my_part * (DM / parts);
  const int pt1 = my_ph * 64 + lane;
  float (*scp)[8] = sc + my_part * (64 * nph);
  if (pt1 < n_slots && my_ph < nph) {
    const float x = px[pt1], y = py[pt1], e = pe[pt1];
    const float* G = w.G;
    const float var = x * (G[0] * x + 2.f * (G[1] * y + G[2] * e + G[3])) + y * (G[4] * y + 2.f * (G[5] * e + G[6])) +
                      e * (G[7] * e + 2.f * G[8]) + G[9];
    const float rstd = 1.0f / sqrtf(fmaxf(var, 0.f) + 1e-5f);
    const f32x2 X = splat2(x), Y = splat2(y), E = splat2(e), RS = splat2(rstd);
    f32x2 s2[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) s2[h] = my_part == 0 ? f32x2{w.cb[2 * h], w.cb[2 * h + 1]} : f32x2{0.f, 0.f};
    auto channels = [&](auto NC) {
      const f32x2* Wc2 = reinterpret_cast<const f32x2*>(w.Wc2) + c_lo * 2;       // [c / 2][4] pairs
      const f32x2* lb2 = reinterpret_cast<const f32x2*>(w.ln_b + c_lo);
      const f32x2* U2 = reinterpret_cast<const f32x2*>(w.U + c_lo * 8);          // [c][4] pairs
      for (int c2 = 0; c2 < decltype(NC)::value / 2; ++c2) {
        const f32x2 d2 = fma2(Wc2[c2 * 4 + 2], E, fma2(Wc2[c2 * 4 + 1], Y, fma2(Wc2[c2 * 4], X, Wc2[c2 * 4 + 3])));
        const f32x2 t2 = fma2(d2, RS, lb2[c2]);
        const float hv0 = fmaxf(t2.x, 0.f), hv1 = fmaxf(t2.y, 0.f);
        const f32x2 H0 = splat2(hv0), H1 = splat2(hv1);
#pragma unroll
        for (int h = 0; h < 4; ++h) s2[h] = fma2(H0, U2[(2 * c2) * 4 + h], s2[h]);
#pragma unroll
        for (int h = 0; h < 4; ++h) s2[h] = fma2(H1, U2[(2 * c2 + 1) * 4 + h], s2[h]);
      }
    };
    if (parts == 1) channels(std::integral_constant<int, DM>{});
    else if (parts == 2) channels(std::integral_constant<int, DM / 2>{});
    else channels(std::integral_constant<int, DM / 4>{});
    if (my_part == 0) stat[pt1] = rstd;
#pragma unroll
    for (int h = 0; h < 4; ++h) { scp[pt1][2 * h] = s2[h].x; scp[pt1][2 * h + 1] = s2[h].y; }
  }
  if (tid == 0) { q0[0] = 0; q0[1] = n0; q0[2] = base1; q0[3] = end1; }
  __syncthreads();
  // ---- softmax over the points of a polyline, per head (one 16-lane DPP row per (polyline, head) pair, as above)
  {
    const int pair = tid >> 4, sub = tid & 15, g = pair >> 3, hd = pair & 7;
    const int a = g < g_here ? q0[2 * g] : 0, b = g < g_here ? q0[2 * g + 1] : 0;
    float mx = -__builtin_inff();
    for (int p = a + sub; p < b; p += 16) {
      float t = sc[p][hd];
      if (parts > 1) {
        for (int q = 1; q < parts; ++q) t += sc[q * 64 * nph + p][hd];
        sc[p][hd] = t;
      }
      mx = fmaxf(mx, t);
    }
    mx = fmaxf(mx, dpp_f32<0xB1>(mx));
    mx = fmaxf(mx, dpp_f32<0x4E>(mx));
    mx = fmaxf(mx, dpp_f32<0x141>(mx));
    mx = fmaxf(mx, dpp_f32<0x140>(mx));
    float z = 0.f;
    for (int p = a + sub; p < b; p += 16) {
      const float ev = expf(sc[p][hd] - mx);
      sc[p][hd] = ev;
      z += ev;
    }
    z += dpp_f32<0xB1>(z);
    z += dpp_f32<0x4E>(z);
    z += dpp_f32<0x141>(z);
    z += dpp_f32<0x140>(z);
    const float inv = 1.0f / z;
    for (int p = a + sub; p < b; p += 16) sc[p][hd] *= inv;
    if (sub == 0 && g < g_here && (b & 1)) sc[b][hd] = 0.f;           // the pad point of an odd range: weight 0
  }
  __syncthreads();
  // ---- phase 2: thread = channel; pooled[h][c][g] = sum_pt a[pt,h] * h1[pt,c], two points per trip
  {
    const int c = tid;
    const f32x2 W0 = splat2(w.Wc[c * 4]), W1 = splat2(w.Wc[c * 4 + 1]), W2 = splat2(w.Wc[c * 4 + 2]), BB = splat2(w.Wc[c * 4 + 3]),
                BE = splat2(w.ln_b[c]);
    f32x2 acc[2][4];
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int h = 0; h < 4; ++h) acc[g][h] = f32x2{0.f, 0.f};
      if (g < g_here) {
        const int a = q0[2 * g], b = (q0[2 * g + 1] + 1) & ~1;
        for (int p = a; p < b; p += 2) {
          const f32x2 x2 = *reinterpret_cast<const f32x2*>(px + p), y2 = *reinterpret_cast<const f32x2*>(py + p),
                      e2 = *reinterpret_cast<const f32x2*>(pe + p), r2 = *reinterpret_cast<const f32x2*>(stat + p);
          const f32x2 d2 = fma2(W2, e2, fma2(W1, y2, fma2(W0, x2, BB)));
          const f32x2 t2 = fma2(d2, r2, BE);
          const f32x2 H0 = splat2(fmaxf(t2.x, 0.f)), H1 = splat2(fmaxf(t2.y, 0.f));
          const f32x2* s0 = reinterpret_cast<const f32x2*>(sc[p]);
          const f32x2* s1 = reinterpret_cast<const f32x2*>(sc[p + 1]);
#pragma unroll
          for (int h = 0; h < 4; ++h) acc[g][h] = fma2(s0[h], H0, acc[g][h]);
#pragma unroll
          for (int h = 0; h < 4; ++h) acc[g][h] = fma2(s1[h], H1, acc[g][h]);
        }
      }
    }
    __syncthreads();                       // every read of the point arrays / sc is done: `pooled` may overwrite them
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      *reinterpret_cast<f32x2*>(pooled[2 * h][c]) = f32x2{acc[0][h].x, acc[1][h].x};
      *reinterpret_cast<f32x2*>(pooled[2 * h + 1][c]) = f32x2{acc[0][h].y, acc[1][h].y};
    }
  }
  __syncthreads();
  // ---- phase 3: thread = output channel j of head j >> 5; the two polylines are the two halves
  {
    const int j = tid, h = j >> 5;
    f32x2 o2 = splat2(w.mb[j]);
#pragma unroll 8
    for (int c = 0; c < DM; ++c) o2 = fma2(*reinterpret_cast<const f32x2*>(pooled[h][c]), splat2(w.Mt[c * DM + j]), o2);
    attn_pre[(size_t)bp0 * DM + j] = o2.x;
    if (g_here > 1) attn_pre[(size_t)(bp0 + 1) * DM + j] = o2.y;
  }
  if (tid < g_here) {
    const int bp = bp0 + tid;
    int k = 0;
    while (k + 1 < mc.n && bp >= mc.bp0[k + 1]) ++k;
    const int rel = bp - mc.bp0[k], b = rel / P, p = rel - b * P;
    src_pad[mc.pad0[k] + (size_t)b * mc.M[k] + p] = (any_exist[tid] && !w.force_pad) ? 0 : 1;
  }
}

// (Round 3 also built this function on the matrix pipe — operand split in the k-slots of v_mfma_f32_16x16x32_f16, both register layouts of
// the hidden tile from swapped operands — correct to 3e-6 and 12 % slower than the kernel above: dependent MFMA -> relu / split -> MFMA
// chains at two workgroups per CU.  Removed in round 4; numbers in profiles/README.md, the code in the history at 8eba46e.)

// n classes: B[k] contexts, M[k] scene rows per context, padding rows of class k from src_pad + pad0[k]; road_pts / attn_pre
// hold the classes back to back
int launch_map_pool_classes(int n, const int* B, const int* M, const long* pad0, int P, int NP, const float* road_pts,
                            MapPoolWeights w, float* attn_pre, unsigned char* src_pad, hipStream_t st) {
  if (n < 1 || n > MAXC || NP < 1 || NP > MAXNP) return CTRLSIM_EINVAL;
  MapClasses mc;
  mc.n = n; mc.bp0[0] = 0;
  for (int k = 0; k < n; ++k) { mc.bp0[k + 1] = mc.bp0[k] + B[k] * P; mc.M[k] = M[k]; mc.pad0[k] = pad0[k]; }
  const int G = NP <= 128 ? 2 : 1, total = mc.bp0[n];
  if (total <= 0) return CTRLSIM_OK;
  prof_before(PROF_MAP, st);
  hipLaunchKernelGGL(MAP_PK ? map_pool_pk_kernel : map_pool_kernel, dim3((total + G - 1) / G), dim3(256), 0, st, NP, P, mc, G, total, road_pts,
                     w, attn_pre, src_pad);
  prof_after(PROF_MAP, 1.5e6 * (double)total, st, (double)total * (12.0 * NP + 4.0 * DM + 1.0));
  return ctrlsim_launch_status();
}
int launch_map_pool(int B, int P, int NP, int M, const float* road_pts, MapPoolWeights w, float* attn_pre,
                    unsigned char* src_pad, hipStream_t st) {
  if (B * P <= 0) return CTRLSIM_OK;
  if (NP < 1 || NP > MAXNP) return CTRLSIM_EINVAL;
  const int G = NP <= 128 ? 2 : 1, total = B * P;
  MapClasses mc;
  mc.n = 1; mc.bp0[0] = 0; mc.bp0[1] = total; mc.M[0] = M; mc.pad0[0] = 0;
  prof_before(PROF_MAP, st);
  hipLaunchKernelGGL(MAP_PK ? map_pool_pk_kernel : map_pool_kernel, dim3((total + G - 1) / G), dim3(256), 0, st, NP, P, mc, G, total, road_pts,
                     w, attn_pre, src_pad);
  // per polyline: NP points x 12 B in, one 256-float row + a padding byte out; ~1.5 MFLOP of folded point MLP + seed attention
  prof_after(PROF_MAP, 1.5e6 * (double)total, st, (double)total * (12.0 * NP + 4.0 * DM + 1.0));
  return ctrlsim_launch_status();
}
