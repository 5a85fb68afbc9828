// Token assembly for the decoder sequence and the scene-encoder's initial-state tokens (gfx950).
//
// Reference: modules/encoder.py:95-153.  Per (timestep tt, agent slot a) three tokens are produced in the order
// (state, rtg, action), sequence index l = (tt*A + a)*3 + k, each  LN_embed( (content + embed_timestep[ts] +
// embed_agent_id[a]) * existence ):
//   state   content = embed_state_goal([embed_state(states||types), embed_goal(goal)])       (:103, :106)
//   rtg     content = embed_rtg([E_goal[b0], E_veh[b1], E_road[b2]])                          (:116-125)
//   action  content = embed_action[token]                                                     (:111)
// Linear algebra that does not depend on the data is folded at weight-pack time (ctrlsim_amd/pack.py, float64):
//   * embed_state.mlp.3 followed by the state half of embed_state_goal is ONE 256x256 matrix; the goal branch
//     likewise; so the state content arrives here as  S2[b,tt,a] (GEMM over the hidden state features) + Gp[b,a].
//   * embed_rtg applied to three embedding rows is the sum of three pre-multiplied 350x256 tables.
// This kernel is the fused gather + add + mask + LayerNorm: one wavefront per (b, tt, a), 4 channels per lane, rows
// written as 1 KiB contiguous stores.  The pre-LN masked state row of window index 0 is also written to the scene
// encoder's source buffer as that agent's "initial state" token (encoder.py:108-109,137-139), with its padding flag.
#include "common.h"

__device__ __forceinline__ f32x4 ln256(f32x4 v, const f32x4 g, const f32x4 b) {
  const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
  const f32x4 d = v - mean;
  const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256.f);
  return d * (1.0f / sqrtf(var + 1e-5f)) * g + b;
}

struct EmbedTables {
  const float* act;        // [V,256]      encoder.embed_action.weight
  const float* rtg_g;      // [R,256]      E_goal @ W_rtg[:, 0:256]^T   (folded)
  const float* rtg_v;      // [R,256]
  const float* rtg_r;      // [R,256]
  const float* rtg_bias;   // [256]
  const float* tstep;      // [MAXT,256]   encoder.embed_timestep.weight
  const float* agent;      // [A,256]      encoder.embed_agent_id.weight
  const float* ln_g;       // [256]        encoder.embed_ln
  const float* ln_b;
  int rtg_linear;          // Decision Transformer: the RTGs are continuous (float bits in rtg_bin) and rtg_g/v/r are single
                           // rows: embed_rtg(cat_c Linear_c(r_c)) = r_0 g + r_1 v + r_2 r + rtg_bias (pack.py fold)
  int flags;               // ctrlsim_dims.flags (include/ctrlsim.h): bit 0 = cfg.model.no_actions — the action embeddings (with their timestep
                           // and agent-id parts) are multiplied by zero before embed_ln (modules/encoder.py:129-130): an action row is
                           // LayerNorm(0) = the norm's bias; bit 2 = encode_initial_state False — the vehicles' initial-state rows are no
                           // keys of the scene encoder / the decoder's memory (modules/encoder.py:159-166): their padding byte is always 1
};

// Compact contexts (forward.hip): the context tensors hold A slots per step, of which the first Areg are "regular" and — when
// Areg < A — the last one is the representative of the padded slots; a context's L token rows are the regular ones,
// (tt*Areg + a)*3 + k, followed from row Lreg on by the representative's, Lreg + 3*tt + k.  Areg == A: the plain layout.
// Classes of contexts in one launch: rows [row0[k], row0[k+1]) of the flat (context, step, slot) list belong to class k with
// A[k] slots (Areg[k] regular), L[k] / Lreg[k] token rows and M[k] scene rows per context; its token rows start at X row
// xrow[k], its scene rows at srow[k], its (context, slot) goal rows at grow[k], its (context, step) timestep entries at trow[k].
// The per-(context, step, slot) inputs (S2, exist, act_tok, rtg_bin) are indexed by the flat row.
struct AsmClasses { int n; int row0[MAXC + 1]; int A[MAXC], Areg[MAXC], L[MAXC], Lreg[MAXC], M[MAXC]; long xrow[MAXC], srow[MAXC], grow[MAXC], trow[MAXC]; };

__global__ __launch_bounds__(256) void assemble_tokens_classes_kernel(
    AsmClasses ac, int Tq, const float* __restrict__ S2, const float* __restrict__ Gp, const float* __restrict__ exist,
    const int* __restrict__ act_tok, const int* __restrict__ rtg_bin, const int* __restrict__ tstep, EmbedTables tb,
    float* __restrict__ X, float* __restrict__ src, int P, unsigned char* __restrict__ src_pad) {
  const int grow_ = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (grow_ >= ac.row0[ac.n]) return;
  int k = 0;
  while (k + 1 < ac.n && grow_ >= ac.row0[k + 1]) ++k;
  k = __builtin_amdgcn_readfirstlane(k);
  const int A = ac.A[k], Areg = ac.Areg[k], L = ac.L[k], Lreg = ac.Lreg[k], M = ac.M[k];
  const int row = grow_ - ac.row0[k];                       // (b, tt, a) within the class
  const int lane = threadIdx.x & 63, c4 = lane * 4;
  const int a = row % A, bt = row / A, tt = bt % Tq, b = bt / Tq;
  const float ex = exist[grow_];
  const int ts = tstep[ac.trow[k] + (size_t)b * Tq + tt];
  const f32x4 g = *reinterpret_cast<const f32x4*>(tb.ln_g + c4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(tb.ln_b + c4);
  const f32x4 pos = *reinterpret_cast<const f32x4*>(tb.tstep + (size_t)ts * DM + c4) +
                    *reinterpret_cast<const f32x4*>(tb.agent + (size_t)a * DM + c4);
  float* xo = X + (ac.xrow[k] + (size_t)b * L + (a < Areg ? ((size_t)tt * Areg + a) * 3 : (size_t)Lreg + 3 * tt)) * DM + c4;
  f32x4 v = (*reinterpret_cast<const f32x4*>(S2 + (size_t)grow_ * DM + c4) +
             *reinterpret_cast<const f32x4*>(Gp + (ac.grow[k] + (size_t)b * A + a) * DM + c4) + pos) * ex;
  if (tt == 0) {
    const size_t s0 = ac.srow[k] + (size_t)b * M;
    *reinterpret_cast<f32x4*>(src + (s0 + P + a) * DM + c4) = v;
    if (lane == 0) src_pad[s0 + P + a] = (ex != 0.f && !(tb.flags & 4)) ? 0 : 1;
    if (a == A - 1) {
      for (int e = P + A; e < M; ++e) {
        *reinterpret_cast<f32x4*>(src + (s0 + e) * DM + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lane == 0) src_pad[s0 + e] = 1;
      }
    }
  }
  *reinterpret_cast<f32x4*>(xo) = ln256(v, g, be);
  const int* rb = rtg_bin + (size_t)grow_ * 3;
  if (tb.rtg_linear) {
    v = (*reinterpret_cast<const f32x4*>(tb.rtg_g + c4) * __int_as_float(rb[0]) +
         *reinterpret_cast<const f32x4*>(tb.rtg_v + c4) * __int_as_float(rb[1]) +
         *reinterpret_cast<const f32x4*>(tb.rtg_r + c4) * __int_as_float(rb[2]) +
         *reinterpret_cast<const f32x4*>(tb.rtg_bias + c4) + pos) * ex;
  } else {
    v = (*reinterpret_cast<const f32x4*>(tb.rtg_g + (size_t)rb[0] * DM + c4) +
         *reinterpret_cast<const f32x4*>(tb.rtg_v + (size_t)rb[1] * DM + c4) +
         *reinterpret_cast<const f32x4*>(tb.rtg_r + (size_t)rb[2] * DM + c4) +
         *reinterpret_cast<const f32x4*>(tb.rtg_bias + c4) + pos) * ex;
  }
  *reinterpret_cast<f32x4*>(xo + DM) = ln256(v, g, be);
  v = (*reinterpret_cast<const f32x4*>(tb.act + (size_t)act_tok[grow_] * DM + c4) + pos) * ((tb.flags & 1) ? 0.f : ex);
  *reinterpret_cast<f32x4*>(xo + 2 * DM) = ln256(v, g, be);
}

__global__ __launch_bounds__(256) void assemble_tokens_kernel(
    int rows, int Tq, int A, int Areg, int L, int Lreg,
    const float* __restrict__ S2,   // [B*Tq*A, 256] state content (without goal part)
    const float* __restrict__ Gp,                            // [B*A, 256] goal part + fused biases
    const float* __restrict__ exist, const int* __restrict__ act_tok, const int* __restrict__ rtg_bin,
    const int* __restrict__ tstep, EmbedTables tb, float* __restrict__ X,   // [B, Tq*A*3, 256]
    float* __restrict__ src, int M, int P,                                   // [B, M, 256] scene-encoder source
    unsigned char* __restrict__ src_pad) {                                   // [B, M] 1 = ignore
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, c4 = lane * 4;
  const int a = row % A, bt = row / A, tt = bt % Tq, b = bt / Tq;
  const float ex = exist[row];
  const int ts = tstep[(size_t)b * Tq + tt];
  const f32x4 g = *reinterpret_cast<const f32x4*>(tb.ln_g + c4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(tb.ln_b + c4);
  const f32x4 pos = *reinterpret_cast<const f32x4*>(tb.tstep + (size_t)ts * DM + c4) +
                    *reinterpret_cast<const f32x4*>(tb.agent + (size_t)a * DM + c4);
  float* xo = X + ((size_t)b * L + (a < Areg ? ((size_t)tt * Areg + a) * 3 : (size_t)Lreg + 3 * tt)) * DM + c4;
  // state
  f32x4 v = (*reinterpret_cast<const f32x4*>(S2 + (size_t)row * DM + c4) +
             *reinterpret_cast<const f32x4*>(Gp + ((size_t)b * A + a) * DM + c4) + pos) * ex;
  if (tt == 0) {
    *reinterpret_cast<f32x4*>(src + ((size_t)b * M + P + a) * DM + c4) = v;
    if (lane == 0) src_pad[(size_t)b * M + P + a] = (ex != 0.f && !(tb.flags & 4)) ? 0 : 1;
    if (a == A - 1) {                              // filler rows up to M (forward.hip: M is rounded up to a multiple of 4): zero, key-padded
      for (int e = P + A; e < M; ++e) {
        *reinterpret_cast<f32x4*>(src + ((size_t)b * M + e) * DM + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lane == 0) src_pad[(size_t)b * M + e] = 1;
      }
    }
  }
  *reinterpret_cast<f32x4*>(xo) = ln256(v, g, be);
  // rtg
  const int* rb = rtg_bin + (size_t)row * 3;
  if (tb.rtg_linear) {
    v = (*reinterpret_cast<const f32x4*>(tb.rtg_g + c4) * __int_as_float(rb[0]) +
         *reinterpret_cast<const f32x4*>(tb.rtg_v + c4) * __int_as_float(rb[1]) +
         *reinterpret_cast<const f32x4*>(tb.rtg_r + c4) * __int_as_float(rb[2]) +
         *reinterpret_cast<const f32x4*>(tb.rtg_bias + c4) + pos) * ex;
  } else {
    v = (*reinterpret_cast<const f32x4*>(tb.rtg_g + (size_t)rb[0] * DM + c4) +
         *reinterpret_cast<const f32x4*>(tb.rtg_v + (size_t)rb[1] * DM + c4) +
         *reinterpret_cast<const f32x4*>(tb.rtg_r + (size_t)rb[2] * DM + c4) +
         *reinterpret_cast<const f32x4*>(tb.rtg_bias + c4) + pos) * ex;
  }
  *reinterpret_cast<f32x4*>(xo + DM) = ln256(v, g, be);
  // action
  v = (*reinterpret_cast<const f32x4*>(tb.act + (size_t)act_tok[row] * DM + c4) + pos) * ((tb.flags & 1) ? 0.f : ex);
  *reinterpret_cast<f32x4*>(xo + 2 * DM) = ln256(v, g, be);
}

// Pass 2: only the RTG tokens of the current timestep change (the sampled bins replace the placeholder); rebuild those
// A rows per context into a compact [B*A, 256] buffer.  hist_rtg [S,N,Tmax,3] holds the bins sampled this step.
__global__ __launch_bounds__(256) void assemble_rtg_rows_kernel(int rows, int Ar, int A, int Tq, int ti, int t, int N, int Tmax,   // Ar rows (regular slots) per context of A slots; Tq/ti: rows per context / row of the current step IN THE CONTEXT TENSORS
                                                                const int* __restrict__ ctx_scn,
                                                                const int* __restrict__ slot_gid,
                                                                const int* __restrict__ hist_rtg,
                                                                const float* __restrict__ exist,
                                                                const int* __restrict__ tstep, EmbedTables tb,
                                                                int zr0, int zr1, int zr2, float* __restrict__ Xr) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);   // row = b*A + a
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, c4 = lane * 4;
  const int a = row % Ar, b = row / Ar;
  const int gid = slot_gid[(size_t)b * A + a];
  int b0 = zr0, b1 = zr1, b2 = zr2;
  if (gid >= 0) {
    const int* rb = hist_rtg + (((size_t)ctx_scn[b] * N + gid) * Tmax + t) * 3;
    b0 = rb[0]; b1 = rb[1]; b2 = rb[2];
  }
  const float ex = exist[((size_t)b * Tq + ti) * A + a];
  const int ts = tstep[(size_t)b * Tq + ti];
  const f32x4 g = *reinterpret_cast<const f32x4*>(tb.ln_g + c4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(tb.ln_b + c4);
  const f32x4 pos = *reinterpret_cast<const f32x4*>(tb.tstep + (size_t)ts * DM + c4) +
                    *reinterpret_cast<const f32x4*>(tb.agent + (size_t)a * DM + c4);
  const f32x4 v = (*reinterpret_cast<const f32x4*>(tb.rtg_g + (size_t)b0 * DM + c4) +
                   *reinterpret_cast<const f32x4*>(tb.rtg_v + (size_t)b1 * DM + c4) +
                   *reinterpret_cast<const f32x4*>(tb.rtg_r + (size_t)b2 * DM + c4) +
                   *reinterpret_cast<const f32x4*>(tb.rtg_bias + c4) + pos) * ex;
  *reinterpret_cast<f32x4*>(Xr + (size_t)row * DM + c4) = ln256(v, g, be);
}

// Cached incremental forward: build only the Rn token rows listed in pos_new (the previous step's action tokens, whose
// ids changed from the placeholder to the applied action, and the 3A tokens of the current timestep; each entry names a
// (window row, slot, token type) of the context tensors) into a compact [B*Rn, 256] buffer.  Context tensors hold the
// window rows [tt_first, tt_first + Tn).
__global__ __launch_bounds__(256) void assemble_rows_kernel(int rows, int Rn, int A, int tt_first, int Tn,
                                                            const int* __restrict__ pos_new,
                                                            const float* __restrict__ S2,    // [B*Tn*A, 256]
                                                            const float* __restrict__ Gp,    // [B*A, 256]
                                                            const float* __restrict__ exist, const int* __restrict__ act_tok,
                                                            const int* __restrict__ rtg_bin, const int* __restrict__ tstep,
                                                            EmbedTables tb, float* __restrict__ Xn) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, c4 = lane * 4;
  const int b = row / Rn, j = row - b * Rn;
  const int code = pos_new[j];                    // (to * A + a) * 3 + k in the context tensors' own (A slots per step) layout
  const int to = code / (3 * A), rem = code - to * 3 * A, a = rem / 3, k = rem - a * 3;
  const size_t cr = ((size_t)b * Tn + to) * A + a;          // row in the context tensors
  const float ex = exist[cr];
  const int ts = tstep[(size_t)b * Tn + to];
  const f32x4 g = *reinterpret_cast<const f32x4*>(tb.ln_g + c4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(tb.ln_b + c4);
  const f32x4 posemb = *reinterpret_cast<const f32x4*>(tb.tstep + (size_t)ts * DM + c4) +
                       *reinterpret_cast<const f32x4*>(tb.agent + (size_t)a * DM + c4);
  f32x4 v;
  if (k == 0) {
    v = *reinterpret_cast<const f32x4*>(S2 + cr * DM + c4) + *reinterpret_cast<const f32x4*>(Gp + ((size_t)b * A + a) * DM + c4);
  } else if (k == 1) {
    const int* rb = rtg_bin + cr * 3;
    v = *reinterpret_cast<const f32x4*>(tb.rtg_g + (size_t)rb[0] * DM + c4) +
        *reinterpret_cast<const f32x4*>(tb.rtg_v + (size_t)rb[1] * DM + c4) +
        *reinterpret_cast<const f32x4*>(tb.rtg_r + (size_t)rb[2] * DM + c4) + *reinterpret_cast<const f32x4*>(tb.rtg_bias + c4);
  } else {
    v = *reinterpret_cast<const f32x4*>(tb.act + (size_t)act_tok[cr] * DM + c4);
  }
  v = (v + posemb) * ((k == 2 && (tb.flags & 1)) ? 0.f : ex);
  *reinterpret_cast<f32x4*>(Xn + (size_t)row * DM + c4) = ln256(v, g, be);
}

int launch_assemble_rows(int B, int Rn, int A, int tt_first, int Tn, const int* pos_new, const float* S2, const float* Gp,
                         const float* exist, const int* act_tok, const int* rtg_bin, const int* tstep, EmbedTables tb,
                         float* Xn, hipStream_t st) {
  const int rows = B * Rn;
  if (rows <= 0) return CTRLSIM_OK;
  hipLaunchKernelGGL(assemble_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, rows, Rn, A, tt_first, Tn, pos_new, S2, Gp,
                     exist, act_tok, rtg_bin, tstep, tb, Xn);
  return ctrlsim_launch_status();
}

int launch_assemble_tokens(int B, int Tq, int A, int Areg, const float* S2, const float* Gp, const float* exist,
                           const int* act_tok, const int* rtg_bin, const int* tstep, EmbedTables tb, float* X,
                           float* src, int M, int P, unsigned char* src_pad, hipStream_t st) {
  const int rows = B * Tq * A;
  if (rows <= 0) return CTRLSIM_OK;
  if (Areg < 1 || Areg > A || A - Areg > 1) return CTRLSIM_EINVAL;
  const int Lreg = Tq * Areg * 3, L = Lreg + (A - Areg) * 3 * Tq;
  prof_before(PROF_EMBED, st);
  hipLaunchKernelGGL(assemble_tokens_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, rows, Tq, A, Areg, L, Lreg, S2, Gp, exist, act_tok,
                     rtg_bin, tstep, tb, X, src, M, P, src_pad);
  // per (context, step, agent): the state row in (1 KB), three token rows out (3 KB), 24 B of ids / existence; embedding
  // tables stay cache-resident
  prof_after(PROF_EMBED, 0.0, st, (double)rows * (4.0 * DM * 4.0 + 24.0));
  return ctrlsim_launch_status();
}

// assemble_tokens for n classes whose context tensors (and S2 / Gp / X / src rows) lie back to back: one launch
int launch_assemble_tokens_classes(int n, const int* B, const int* A, const int* Areg, const int* M, const long* xrow,
                                   const long* srow, const long* grow, int Tq, const float* S2,
                                   const float* Gp, const float* exist, const int* act_tok, const int* rtg_bin, const int* tstep,
                                   EmbedTables tb, float* X, float* src, int P, unsigned char* src_pad, hipStream_t st) {
  if (n < 1 || n > MAXC) return CTRLSIM_EINVAL;
  AsmClasses ac;
  ac.n = n; ac.row0[0] = 0;
  long tr = 0;
  for (int k = 0; k < n; ++k) {
    if (Areg[k] < 1 || Areg[k] > A[k] || A[k] - Areg[k] > 1) return CTRLSIM_EINVAL;
    ac.A[k] = A[k]; ac.Areg[k] = Areg[k]; ac.M[k] = M[k];
    ac.Lreg[k] = Tq * Areg[k] * 3; ac.L[k] = ac.Lreg[k] + (A[k] - Areg[k]) * 3 * Tq;
    ac.xrow[k] = xrow[k]; ac.srow[k] = srow[k]; ac.grow[k] = grow[k]; ac.trow[k] = tr;
    ac.row0[k + 1] = ac.row0[k] + B[k] * Tq * A[k];
    tr += (long)B[k] * Tq;
  }
  const int rows = ac.row0[n];
  if (rows <= 0) return CTRLSIM_OK;
  prof_before(PROF_EMBED, st);
  hipLaunchKernelGGL(assemble_tokens_classes_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, ac, Tq, S2, Gp, exist, act_tok, rtg_bin,
                     tstep, tb, X, src, P, src_pad);
  prof_after(PROF_EMBED, 0.0, st, (double)rows * (4.0 * DM * 4.0 + 24.0));
  return ctrlsim_launch_status();
}

int launch_assemble_rtg_rows(int B, int Ar, int A, int Tq, int ti, int t, int N, int Tmax, const int* ctx_scn,
                             const int* slot_gid, const int* hist_rtg, const float* exist, const int* tstep,
                             EmbedTables tb, const int* zr, float* Xr, hipStream_t st) {
  const int rows = B * Ar;
  if (rows <= 0) return CTRLSIM_OK;
  hipLaunchKernelGGL(assemble_rtg_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, rows, Ar, A, Tq, ti, t, N, Tmax, ctx_scn,
                     slot_gid, hist_rtg, exist, tstep, tb, zr[0], zr[1], zr[2], Xr);
  return ctrlsim_launch_status();
}
