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
#define SPLIT_MIX
#define CTRLSIM_F16X3 1      // the MFMA variant below exists for the two-fp16-plane operand split only (split.h helpers)
#include "split.h"

#define MAXNP 256

struct MapPoolWeights {
  const float* Wc;      // [256,4]  g_c * (W1[c,:] - column mean, b1[c] - mean(b1)): LN(W1 p + b1)_c = Wc[c] . (x,y,e,1) * rstd + ln_b[c]
  const float* G;       // [10]     upper triangle of sum_c wt_c wt_c^T / 256 (wt = Wc without the gain): var = (x,y,e,1)^T G (x,y,e,1)
  const float* ln_b;    // [256]
  const float* U;       // [256,8]
  const float* cb;      // [8]
  const float* Mt;      // [256(c),256(j)]
  const float* mb;      // [256]
  const void* wfrag;    // MFMA variant: fp16 operand fragments of (Wc | ln_b)   [16 channel tiles][64 lanes][8]  (pack.py: map_frags)
  const void* ufrag;    //               fp16 operand fragments of 2^8 U, hi / lo [8 k-steps][64 lanes][8]
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
  // ---- phase 1: per visible point LN statistics and head scores
  if (tid < n_vis) {
    const float x = pts[tid][0], y = pts[tid][1], e = pts[tid][2];
    // LayerNorm statistics in closed form (the layer is affine in the point: pack.py); the mean drops out of the centred weights
    const float* G = w.G;
    const float var = x * (G[0] * x + 2.f * (G[1] * y + G[2] * e + G[3])) + y * (G[4] * y + 2.f * (G[5] * e + G[6])) +
                      e * (G[7] * e + 2.f * G[8]) + G[9];
    const float rstd = 1.0f / sqrtf(fmaxf(var, 0.f) + 1e-5f);
    float s8[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) s8[h] = w.cb[h];
#ifdef MPV_ABL_NO_P1
    for (int c = 0; c < (x > 1e30f ? DM : 0); ++c) {
#else
    for (int c = 0; c < DM; ++c) {
#endif
      const float d = fmaf(w.Wc[c * 4 + 2], e, fmaf(w.Wc[c * 4 + 1], y, fmaf(w.Wc[c * 4], x, w.Wc[c * 4 + 3])));
      const float hv = fmaxf(fmaf(d, rstd, w.ln_b[c]), 0.f);
#pragma unroll
      for (int h = 0; h < 8; ++h) s8[h] = fmaf(hv, w.U[c * 8 + h], s8[h]);
    }
    stat[tid] = rstd;
#pragma unroll
    for (int h = 0; h < 8; ++h) sc[tid][h] = s8[h];
  }
  __syncthreads();
  // ---- softmax over the visible points of a polyline, per head
  if (tid < 8 * g_here) {
    const int g = tid >> 3, hd = tid & 7, a = q0[g], b = q0[g + 1];
    float mx = -__builtin_inff();
    for (int p = a; p < b; ++p) mx = fmaxf(mx, sc[p][hd]);
    float z = 0.f;
    for (int p = a; p < b; ++p) {
      const float ev = expf(sc[p][hd] - mx);
      sc[p][hd] = ev;
      z += ev;
    }
    const float inv = 1.0f / z;
    for (int p = a; p < b; ++p) sc[p][hd] *= inv;
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
    src_pad[mc.pad0[k] + (size_t)b * mc.M[k] + p] = any_exist[tid] ? 0 : 1;
  }
}

// ---------------------------------------------------------------------------------------------------- MFMA variant
// The same function on the matrix pipe (two-fp16-plane operand split, fp32 accumulation: fp32-class results).  What makes it fit:
//   * the first layer + LayerNorm is a K = 5 contraction:  LN(.)_c * std = Wc[c].(x, y, e, 1) + ln_b[c] * std =: D[c, pt], and with
//     R = relu(D) the hidden vector is h1 = rstd * R (rstd > 0) — rstd moves out of both products below;
//   * the operand split lives in the K dimension of ONE v_mfma_f32_16x16x32_f16: the 13 used k-slots are the partial products
//     w_hi x_hi, w_lo x_hi, w_hi x_lo (x, y, std), w_hi e, w_lo e, w_hi 1, w_lo 1 — so a 16 x 16 tile of D costs one instruction;
//   * D is produced in BOTH register layouts the two contractions need, by swapping the MFMA operands:  D^T = Wc X^T gives
//     lane = point / registers = channels (the B operand of  scores^T = U^T R, contraction over channels), D = X Wc^T gives
//     lane = channel / registers = points (the B operand of  pooled = (a rstd)^T R, contraction over points); the accumulator-register
//     order IS the k-slot order of the consuming MFMA (constant operands are packed to match: pack.py map_frags), so R never
//     moves between lanes — the trick of the attention kernel's P operand;
//   * hi / lo planes of the small operand ride in the M dimension: rows 0-7 = U_hi (a_hi), rows 8-15 = U_lo (a_lo) of the 8 heads,
//     so the three partial products cost two instructions (B = R_hi, B = R_lo) instead of three.
// Per (point, channel) element: relu + split = 2.5 VALU in each layout instead of 13 + 13; the products themselves: 64 MFMA
// cycles per point and SIMD.  Softmax weights are scaled per (polyline, head) by a power of two that puts max(a rstd) just
// below 2^14 (exact; fp16 holds every term to 2^-22 of the largest) and unscaled after the sum.
// Persistent workgroups (the operand fragments, 24 KB, are loaded into LDS once), two polylines per pass; each polyline's compacted
// points start at a multiple of 32 so that a 32-point k-step of the pooling product never straddles two polylines.
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
#define MP_PTS 256            // compact point slots of a pass (2 polylines x <= 128 points)

__device__ __forceinline__ void relu_split8(const f32x4_ d0, const f32x4_ d1, opx8 (&rf)[NPL]) {
  const float r[8] = {fmaxf(d0[0], 0.f), fmaxf(d0[1], 0.f), fmaxf(d0[2], 0.f), fmaxf(d0[3], 0.f),
                      fmaxf(d1[0], 0.f), fmaxf(d1[1], 0.f), fmaxf(d1[2], 0.f), fmaxf(d1[3], 0.f)};
  split_frag(r, rf);
}

__global__ __launch_bounds__(256, 2) void map_pool_mfma_kernel(int NP, int P, MapClasses mc, int total, const float* __restrict__ road_pts,
                                                               MapPoolWeights w, float* __restrict__ attn_pre,
                                                               unsigned char* __restrict__ src_pad) {
  __shared__ __attribute__((aligned(16))) opx8 cw[16 * 64];                    // (Wc | ln_b) fragments, 16 channel tiles
  __shared__ __attribute__((aligned(16))) opx8 cu[8 * 64];                     // 2^8 U fragments, 8 k-steps of 32 channels
  __shared__ __attribute__((aligned(16))) struct { opx8 xfrag[(MP_PTS / 16) * 64]; float pooled[2][8][DM]; } u;   // 16 KB + 16 KB
  __shared__ __attribute__((aligned(16))) float pts[MP_PTS][4];                // x, y, e, rstd of the compacted points
  __shared__ __attribute__((aligned(16))) float sc[MP_PTS][8];
  __shared__ __attribute__((aligned(16))) _Float16 atab[16][MP_PTS];           // rows 0-7: hi plane of a rstd 2^k per head, 8-15: lo
  __shared__ float inv_scale[2][8];
  __shared__ int any_exist[2], wcnt[4], wc0[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m16 = lane & 15, g4 = lane >> 4;
  {
    const opx8* wg = static_cast<const opx8*>(w.wfrag);
    const opx8* ug = static_cast<const opx8*>(w.ufrag);
    for (int i = tid; i < 16 * 64; i += 256) cw[i] = wg[i];
    for (int i = tid; i < 8 * 64; i += 256) cu[i] = ug[i];
  }
  {
    // the point fragments are cleared ONCE: slot groups 2 / 3 stay zero for good, and entries left over from an earlier pass (points
    // beyond this pass's ranges) are finite and meet zero weights in the pooling product, unread scores in the first
    const opx8 z = {};
    for (int i = tid; i < (MP_PTS / 16) * 64; i += 256) u.xfrag[i] = z;
  }
  const int n_pairs = (total + 1) >> 1;
  for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
    const int bp0 = pair * 2;
    const int g_here = min(2, total - bp0);
    const int n_pts = g_here * NP;
    const float* src = road_pts + (size_t)bp0 * NP * 3;
    // ---- (a) clear the per-pass operand tables
    {
      const opx8 z = {};
      opx8* at = reinterpret_cast<opx8*>(&atab[0][0]);
      for (int i = tid; i < 16 * MP_PTS / 8; i += 256) at[i] = z;
      if (tid < 2) any_exist[tid] = 0;
    }
    __syncthreads();
    // ---- (b) points, key padding (map_encoder.py:31), compaction: polyline 0 from slot 0, polyline 1 from the next multiple of 32
    const int g_of = tid >= NP, p_in = tid - g_of * NP;
    float x = 0.f, y = 0.f, e = 0.f;
    if (tid < n_pts) {
      x = src[tid * 3]; y = src[tid * 3 + 1]; e = src[tid * 3 + 2];
      if (e != 0.f) any_exist[g_of] = 1;
    }
    __syncthreads();
    const bool vis = tid < n_pts && (e != 0.f || (any_exist[g_of] == 0 && p_in == 0));
    const unsigned long long bal = __ballot(vis);
    {
      const int first1 = NP - wv * 64;                          // lanes >= first1 of this wave belong to polyline 1
      const unsigned long long m0 = first1 >= 64 ? ~0ull : (first1 <= 0 ? 0ull : ((1ull << first1) - 1ull));
      if (lane == 0) { wcnt[wv] = __popcll(bal); wc0[wv] = __popcll(bal & m0); }
    }
    __syncthreads();
    int before = __popcll(bal & ((1ull << lane) - 1ull));
    for (int k = 0; k < wv; ++k) before += wcnt[k];
    const int cnt0 = wc0[0] + wc0[1] + wc0[2] + wc0[3], n_all = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    const int e0 = cnt0, q1 = (cnt0 + 31) & ~31, e1 = q1 + (n_all - cnt0);      // polyline 0: [0, e0), polyline 1: [q1, e1)
    if (vis) {
      const int idx = g_of ? q1 + (before - cnt0) : before;
      pts[idx][0] = x; pts[idx][1] = y; pts[idx][2] = e;
    }
    __syncthreads();
    // ---- (c) per point: LayerNorm statistics in closed form, operand fragment of (x, y, e, 1, std) with its partial-product slots
    if (tid < e1 && (tid < e0 || tid >= q1)) {
      const float px = pts[tid][0], py = pts[tid][1], pe = pts[tid][2];
      const float* G = w.G;
      const float var = px * (G[0] * px + 2.f * (G[1] * py + G[2] * pe + G[3])) + py * (G[4] * py + 2.f * (G[5] * pe + G[6])) +
                        pe * (G[7] * pe + 2.f * G[8]) + G[9];
      const float sd = sqrtf(fmaxf(var, 0.f) + 1e-5f);
      pts[tid][3] = 1.0f / sd;
      unsigned xs[NPL], ss[NPL];
      split_pair(px, py, xs);                                   // low half: x planes, high half: y planes
      split_pair(sd, 0.f, ss);
      const unsigned xh = xs[0] & 0xffffu, xl = xs[1] & 0xffffu, yh = xs[0] >> 16, yl = xs[1] >> 16;
      const unsigned sh = ss[0] & 0xffffu, sl = ss[1] & 0xffffu;
      const unsigned eh = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)pe), one = 0x3c00u;
      // slots (pairing with pack.py map_frags):  g0: x_hi x_hi x_lo y_hi y_hi y_lo e e     g1: 1 1 s_hi s_hi s_lo 0 0 0
      const u32x4 f0 = {xh | (xh << 16), xl | (yh << 16), yh | (yl << 16), eh | (eh << 16)};
      const u32x4 f1 = {one | (one << 16), sh | (sh << 16), sl, 0u};
      const int t = tid >> 4, mm = tid & 15;
      u.xfrag[t * 64 + mm] = __builtin_bit_cast(opx8, f0);
      u.xfrag[t * 64 + 16 + mm] = __builtin_bit_cast(opx8, f1);
    }
    __syncthreads();
#ifndef MP_ABL_NO_P1
    // ---- (d) phase 1: scores.  A wave takes 16-point tiles; per k-step of 32 channels: two D^T tiles, relu, split, two score MFMAs
    const int n_tiles = (e1 + 15) >> 4;
    // (two tiles per trip: two independent MFMA -> relu / split -> MFMA chains in flight per wave; a tile index past the end
    // computes on left-over fragments and is not stored)
    for (int t = wv; t < n_tiles; t += 8) {
      const int t2 = t + 4;
      const opx8 xf = u.xfrag[t * 64 + lane], xg = u.xfrag[(t2 < MP_PTS / 16 ? t2 : t) * 64 + lane];
      f32x4_ accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f}, accC = {0.f, 0.f, 0.f, 0.f}, accD = {0.f, 0.f, 0.f, 0.f};
      const f32x4_ z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const opx8 w0 = cw[(2 * s) * 64 + lane], w1 = cw[(2 * s + 1) * 64 + lane], uf = cu[s * 64 + lane];
        const f32x4_ d0 = MFMA16(w0, xf, z4), d1 = MFMA16(w1, xf, z4);
        const f32x4_ e0_ = MFMA16(w0, xg, z4), e1_ = MFMA16(w1, xg, z4);
        opx8 rf[NPL], rg[NPL];
        relu_split8(d0, d1, rf);
        relu_split8(e0_, e1_, rg);
        accA = MFMA16(uf, rf[0], accA);                          // rows 0-7: U_hi R_hi, rows 8-15: U_lo R_hi
        accB = MFMA16(uf, rf[1], accB);                          // rows 0-7: U_hi R_lo
        accC = MFMA16(uf, rg[0], accC);
        accD = MFMA16(uf, rg[1], accD);
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f32x4_ qa = half ? accC : accA, qb = half ? accD : accB;
        float oth[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) oth[r] = __shfl_xor(qa[r], 32);    // the U_lo rows of the same heads live 32 lanes up
        const int p = (half ? t2 : t) * 16 + m16;
        if (g4 < 2 && p < e1) {
          const float rs = pts[p][3] * 0.00390625f;             // rstd / 2^8 (the U fragments carry 2^8)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[p][4 * g4 + r] = fmaf(rs, (qa[r] + qb[r]) + oth[r], w.cb[4 * g4 + r]);
        }
      }
    }
#endif
    __syncthreads();
    // ---- (e) softmax over the points of a polyline per head; a rstd scaled into the fp16 planes of the pooling operand
    for (int id = wv * 4; id < wv * 4 + 4; ++id) {
      const int gi = id >> 3, hd = id & 7;
      if (gi >= g_here) continue;
      const int a = gi ? q1 : 0, b = gi ? e1 : e0;
      const int p0 = a + lane, p1 = a + 64 + lane;
      const float v0 = p0 < b ? sc[p0][hd] : -__builtin_inff(), v1 = p1 < b ? sc[p1][hd] : -__builtin_inff();
      const float mx = wave_max(fmaxf(v0, v1));
      const float e0v = p0 < b ? expf(v0 - mx) : 0.f, e1v = p1 < b ? expf(v1 - mx) : 0.f;
      const float inv = 1.0f / wave_sum(e0v + e1v);
      const float ar0 = p0 < b ? e0v * inv * pts[p0][3] : 0.f, ar1 = p1 < b ? e1v * inv * pts[p1][3] : 0.f;
      const float mxar = wave_max(fmaxf(ar0, ar1));
      int ex = 0;
      (void)frexpf(mxar, &ex);                                   // mxar = m 2^ex, m in [0.5, 1)
      const float scl = ldexpf(1.0f, 14 - ex);                   // max(a rstd) 2^k in [2^13, 2^14)
      if (lane == 0) inv_scale[gi][hd] = ldexpf(1.0f, ex - 14);
      unsigned pl[NPL];
      split_pair(ar0 * scl, ar1 * scl, pl);
      if (p0 < b) {
        atab[hd][p0] = __builtin_bit_cast(_Float16, (unsigned short)(pl[0] & 0xffffu));
        atab[8 + hd][p0] = __builtin_bit_cast(_Float16, (unsigned short)(pl[1] & 0xffffu));
      }
      if (p1 < b) {
        atab[hd][p1] = __builtin_bit_cast(_Float16, (unsigned short)(pl[0] >> 16));
        atab[8 + hd][p1] = __builtin_bit_cast(_Float16, (unsigned short)(pl[1] >> 16));
      }
    }
    __syncthreads();
    // ---- (f) phase 2: pooled[g][h][c] = sum_pt (a rstd)[pt, h] R[pt, c].  A wave owns 4 channel tiles; k-steps of 32 points
    f32x4_ pa[2][4], pb[2][4];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) { pa[gi][ct] = f32x4_{0.f, 0.f, 0.f, 0.f}; pb[gi][ct] = f32x4_{0.f, 0.f, 0.f, 0.f}; }
#ifndef MP_ABL_NO_P2
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      if (gi >= g_here) continue;
      const int ks0 = gi ? (q1 >> 5) : 0, ks1 = ((gi ? e1 : e0) + 31) >> 5;
      for (int ks = ks0; ks < ks1; ++ks) {
        const opx8 xf0 = u.xfrag[(2 * ks) * 64 + lane], xf1 = u.xfrag[(2 * ks + 1) * 64 + lane];
        const h4 a_lo = *reinterpret_cast<const h4*>(&atab[m16][32 * ks + 4 * g4]);
        const h4 a_hi = *reinterpret_cast<const h4*>(&atab[m16][32 * ks + 16 + 4 * g4]);
        const opx8 af = __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7);
        const f32x4_ z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const opx8 wf = cw[(4 * wv + ct) * 64 + lane];
          const f32x4_ d0 = MFMA16(xf0, wf, z4);                 // [16 points x 16 channels]: lane = channel, registers = points
          const f32x4_ d1 = MFMA16(xf1, wf, z4);
          opx8 rf[NPL];
          relu_split8(d0, d1, rf);
          pa[gi][ct] = MFMA16(af, rf[0], pa[gi][ct]);
          pb[gi][ct] = MFMA16(af, rf[1], pb[gi][ct]);
        }
      }
    }
#endif
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        float oth[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) oth[r] = __shfl_xor(pa[gi][ct][r], 32);
        if (g4 < 2) {
          const int c = 16 * (4 * wv + ct) + m16;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            u.pooled[gi][4 * g4 + r][c] = ((pa[gi][ct][r] + pb[gi][ct][r]) + oth[r]) * inv_scale[gi][4 * g4 + r];
        }
      }
    __syncthreads();
    // ---- (g) phase 3: thread = output channel j of head j >> 5 (the folded 256 x 256 matrix once for both polylines)
    {
      const int j = tid, h = j >> 5;
      float o0 = w.mb[j], o1 = o0;
#ifndef MP_ABL_NO_P3
#pragma unroll 32
      for (int c = 0; c < DM; ++c) {
        const float mm = w.Mt[c * DM + j];
        o0 = fmaf(u.pooled[0][h][c], mm, o0);
        o1 = fmaf(u.pooled[1][h][c], mm, o1);
      }
#else
      o0 += u.pooled[0][h][j]; o1 += u.pooled[1][h][j];
#endif
      attn_pre[(size_t)bp0 * DM + j] = o0;
      if (g_here > 1) attn_pre[(size_t)(bp0 + 1) * DM + j] = o1;
    }
    if (tid < g_here) {
      const int bp = bp0 + tid;
      int k = 0;
      while (k + 1 < mc.n && bp >= mc.bp0[k + 1]) ++k;
      const int rel = bp - mc.bp0[k], bq = rel / P, pq = rel - bq * P;
      src_pad[mc.pad0[k] + (size_t)bq * mc.M[k] + pq] = any_exist[tid] ? 0 : 1;
    }
    __syncthreads();                                             // the next pass clears xfrag (= pooled) and any_exist
  }
}

static bool map_pool_use_mfma(int NP, const MapPoolWeights& w) {
  return ctrlsim_option(OPT_MAP_MFMA) == 1 && ctrlsim_option(OPT_SPLIT) == 1 && NP <= 128 && w.wfrag && w.ufrag;
}
static int map_pool_mfma_grid(int total) {
  static const int n_cus = [] {
    int dev = 0, n = 0;
    return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }();
  const int n_pairs = (total + 1) / 2;
  return n_pairs < 2 * n_cus ? n_pairs : 2 * n_cus;             // two resident workgroups per CU (60 KB of LDS each)
}

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
  if (map_pool_use_mfma(NP, w))
    hipLaunchKernelGGL(map_pool_mfma_kernel, dim3(map_pool_mfma_grid(total)), dim3(256), 0, st, NP, P, mc, total, road_pts, w, attn_pre, src_pad);
  else
  hipLaunchKernelGGL(map_pool_kernel, dim3((total + G - 1) / G), dim3(256), 0, st, NP, P, mc, G, total, road_pts, w, attn_pre,
                     src_pad);
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
  if (map_pool_use_mfma(NP, w))
    hipLaunchKernelGGL(map_pool_mfma_kernel, dim3(map_pool_mfma_grid(total)), dim3(256), 0, st, NP, P, mc, total, road_pts, w, attn_pre, src_pad);
  else
  hipLaunchKernelGGL(map_pool_kernel, dim3((total + G - 1) / G), dim3(256), 0, st, NP, P, mc, G, total, road_pts, w, attn_pre,
                     src_pad);
  // per polyline: NP points x 12 B in, one 256-float row + a padding byte out; ~1.5 MFLOP of folded point MLP + seed attention
  prof_after(PROF_MAP, 1.5e6 * (double)total, st, (double)total * (12.0 * NP + 4.0 * DM + 1.0));
  return ctrlsim_launch_status();
}
