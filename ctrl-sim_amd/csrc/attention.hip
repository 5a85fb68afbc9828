// Streaming (flash-style) multi-head attention in fp32 on the f32-input MFMA, head_dim 32, gfx950.
//
// Serves every attention of the reference model (all torch.nn.MultiheadAttention, 8 heads x 32, fp32):
//   MODE_CAUSAL  decoder self-attention, modules/decoder.py:52 with the additive mask of
//                utils/train_utils.py:81-129 evaluated arithmetically — token i = (t*A + a)*3 + k sees token j iff
//                t_j < t_i  or  (t_j == t_i and ((a_j == a_i and k_j <= k_i) or k_j == 0))
//                (closed form proven equal to get_causal_mask for the CtRL-Sim variant).
//                No 2304^2 mask ever exists in HBM; key tiles entirely in the future are skipped, tiles entirely
//                in the past skip the mask arithmetic.
//   MODE_KEYPAD  scene-encoder self-attention (src_key_padding_mask, modules/encoder.py:155-168) and decoder
//                cross-attention (memory_key_padding_mask, modules/decoder.py:52): a per-key boolean, no causality.
//
// Work split: one workgroup = 4 waves = 128 queries of one (context, head); K/V tiles of 64 keys are staged in LDS
// (rows padded to 36 floats) and shared by the 4 waves.  Per wave and 32-key sub-tile:
//   S^T = K.Q^T   : 16 x v_mfma_f32_32x32x2_f32 (A = K rows from LDS, B = the wave's Q fragment held in registers).
//                   The transposed product puts ONE query per lane column (q = lane&31), so the running max / sum
//                   of the online softmax are lane-local plus one exchange with lane^32.
//   O^T += V^T.P^T: 16 more MFMAs; the exponentiated S^T accumulator registers are used directly as the B operand
//                   (their (register, lane-half) -> key map is the MFMA k-slot map), V rows are read from LDS as A.
// The rescale of O^T by exp(m_old - m_new) is lane-local for the same reason.  Output is transposed through LDS so
// that global stores are 128-byte rows.
//
// Queries may be a gathered subset (q_pos != nullptr): Q/O are then compact [B*Lq, .] buffers and q_pos[i] gives the
// token position used by the mask (pass-2 / last-layer evaluation of the current-timestep tokens only).
#include "common.h"

#ifndef KT
#define KT 64
#endif
#ifndef ATT_WPS
#define ATT_WPS 4
#endif
#define KP 36

enum { MODE_KEYPAD = 0, MODE_CAUSAL = 1 };

template <int MODE>
__global__ __launch_bounds__(256, ATT_WPS) void attention_f32_kernel(
    const float* __restrict__ Q, int ldq, long q_batch_stride,   // Q[b*q_batch_stride + i*ldq + h*32 + d]
    const float* __restrict__ K, const float* __restrict__ V, int ldkv, long kv_batch_stride,
    float* __restrict__ O, int ldo, long o_batch_stride, const int* __restrict__ q_pos,
    const unsigned char* __restrict__ key_pad,                    // [B, Lk], 1 = ignore (MODE_KEYPAD)
    int Lq, int Lk, int A, float scale_log2e) {
  // one LDS arena: K/V staging during the key loop, re-used for the output transpose afterwards
  // (two K/V buffers: the next tile is written while the current one is read, one barrier per tile)
  __shared__ __attribute__((aligned(16))) float arena[2 * (2 * KT * KP + KT)];
  __shared__ int blk_tmax[4];
  constexpr int BUF = 2 * KT * KP + KT;

  // causal blocks are issued heaviest (latest queries) first so the tail of the launch is made of short blocks
  const int qblk = (MODE == MODE_CAUSAL) ? (gridDim.x - 1 - blockIdx.x) : blockIdx.x;
  const int b = blockIdx.z, h = blockIdx.y, qb = qblk * 128;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int A3 = 3 * A;
  const float NEG_INF = -__builtin_inff();

  // ---- this lane's query
  const int qi = qb + wave * 32 + l31;
  const bool qvalid = qi < Lq;
  const int qrow = qvalid ? qi : (Lq - 1);
  const int pos = q_pos ? q_pos[qrow] : qrow;
  int tq = 0, aq = 0, kq = 0;
  if (MODE == MODE_CAUSAL) {
    tq = pos / A3;
    const int rem = pos - tq * A3;
    aq = rem / 3;
    kq = rem - aq * 3;
  }
  const float* qp = Q + (size_t)b * q_batch_stride + (size_t)qrow * ldq + h * HD + half * 4;
  f32x4 qf[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    qf[j] = *reinterpret_cast<const f32x4*>(qp + j * 8);
    qf[j] *= scale_log2e;                                   // scores live in the log2 domain: p = exp2(s - m)
  }

  // ---- key range
  int k_end = Lk;
  int tq_min_w = 0, tq_max_w = 0;
  if (MODE == MODE_CAUSAL) {
    int tmin = tq, tmax = tq;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      tmin = min(tmin, __shfl_xor(tmin, o, 64));
      tmax = max(tmax, __shfl_xor(tmax, o, 64));
    }
    tq_min_w = tmin;
    tq_max_w = tmax;
    if (lane == 0) blk_tmax[wave] = tmax;
    __syncthreads();
    const int bt = max(max(blk_tmax[0], blk_tmax[1]), max(blk_tmax[2], blk_tmax[3]));
    k_end = min(Lk, (bt + 1) * A3);
  }

  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float m_run = NEG_INF, l_run = 0.f;

  const float* Kb = K + (size_t)b * kv_batch_stride + h * HD;
  const float* Vb = V + (size_t)b * kv_batch_stride + h * HD;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // register prefetch of the next K/V tile (KT keys x 32 dims each: KT/32 float4 per thread and operand)
  constexpr int LDI = KT / 32;
  f32x4 pk[LDI], pv[LDI];
  float ppad = 0.f;
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LDI; ++i) {
      const int idx = tid + 256 * i, r = idx >> 3, c = (idx & 7) * 4;
      const int kr = k0 + r;
      pk[i] = zero4; pv[i] = zero4;
      if (kr < Lk) {
        pk[i] = *reinterpret_cast<const f32x4*>(Kb + (size_t)kr * ldkv + c);
        pv[i] = *reinterpret_cast<const f32x4*>(Vb + (size_t)kr * ldkv + c);
      }
    }
    if (MODE == MODE_KEYPAD && tid < KT) {
      const int kr = k0 + tid;
      ppad = (kr < Lk && !key_pad[(size_t)b * Lk + kr]) ? 0.f : NEG_INF;
    }
  };
  auto sstore = [&](int buf) {
    float* Kd = arena + buf * BUF;
    float* Vd = Kd + KT * KP;
#pragma unroll
    for (int i = 0; i < LDI; ++i) {
      const int idx = tid + 256 * i, r = idx >> 3, c = (idx & 7) * 4;
      *reinterpret_cast<f32x4*>(Kd + r * KP + c) = pk[i];
      *reinterpret_cast<f32x4*>(Vd + r * KP + c) = pv[i];
    }
    if (MODE == MODE_KEYPAD && tid < KT) Kd[2 * KT * KP + tid] = ppad;
  };

  if (k_end > 0) {
    gload(0);
    sstore(0);
  }
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < k_end; k0 += KT, cur ^= 1) {
    const bool more = k0 + KT < k_end;
    if (more) gload(k0 + KT);                               // in flight while this tile is consumed
    const float* Ks = arena + cur * BUF;
    const float* Vs = Ks + KT * KP;
    const float* padbias = Ks + 2 * KT * KP;

#pragma unroll
    for (int sub = 0; sub < KT / 32; ++sub) {
      const int ks = k0 + sub * 32;  // first key of the sub-tile
      if (ks >= k_end) continue;
      bool need_mask = true;
      if (MODE == MODE_CAUSAL) {
        const int t_lo = ks / A3, t_hi = min(ks + 31, Lk - 1) / A3;
        if (t_lo > tq_max_w) continue;                      // every key is in the future of every query of this wave
        need_mask = !(t_hi < tq_min_w && ks + 31 < Lk);     // strictly in the past: all visible
      }
      // ---- S^T = K . Q^T
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      const float* kr_ = Ks + (sub * 32 + l31) * KP + half * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(kr_ + j * 8);
#pragma unroll
        for (int c = 0; c < 4; ++c) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c], qf[j][c], s, 0, 0, 0);
      }
      // ---- mask into a separate VGPR array (the MFMA accumulator itself is never edited in place: hipcc 7.2 was seen
      //      to lose in-place element writes to an AGPR-resident accumulator across the masked/unmasked join)
      float sc[16];
      if (MODE == MODE_KEYPAD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = s[r] + padbias[sub * 32 + mfma_row(r, half)];
      } else if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kj = ks + mfma_row(r, half);
          const int tj = kj / A3;
          const int rem = kj - tj * A3;
          const int aj = rem / 3;
          const int kk = rem - aj * 3;
          const bool vis = (kj < Lk) && ((tj < tq) || (tj == tq && ((aj == aq && kk <= kq) || kk == 0)));
          sc[r] = vis ? s[r] : NEG_INF;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = s[r];
      }
      // ---- online softmax (one query per lane column; partner lane^32 holds the other 16 keys)
      float tmax = sc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax);
      const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);   // m_run = -inf -> 0
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sc[r] = __builtin_amdgcn_exp2f(sc[r] - m_use);
        psum += sc[r];
      }
      psum += __shfl_xor(psum, 32, 64);
      l_run = l_run * alpha + psum;
      m_run = m_new;
      if (!__all(alpha == 1.0f)) {                         // running max unchanged for the whole wave: no rescale
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
      }
      // ---- O^T += V^T . P^T
      const float* vr_ = Vs + (sub * 32) * KP + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float vf = vr_[mfma_row(r, half) * KP];
        oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, sc[r], oacc, 0, 0, 0);
      }
    }
    if (more) sstore(cur ^ 1);                              // the other buffer was last read before the previous barrier
    __syncthreads();
  }

  // ---- normalise, transpose through LDS (arena is free: every wave passed the loop's final barrier), store rows
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  float* ot = arena + wave * (32 * 33);
#pragma unroll
  for (int r = 0; r < 16; ++r) ot[l31 * 33 + mfma_row(r, half)] = oacc[r] * inv;   // ot[q][d]
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int q = i * 2 + half;
    const int gq = qb + wave * 32 + q;
    if (gq < Lq) O[(size_t)b * o_batch_stride + (size_t)gq * ldo + h * HD + l31] = ot[q * 33 + l31];
  }
}

int launch_attention_bf16x6(int, const float*, int, long, const float*, const float*, int, long, float*, int, long, const int*,
                            const unsigned char*, int, int, int, int, hipStream_t);

int launch_attention(int mode, const float* Q, int ldq, long q_batch_stride, const float* K, const float* V, int ldkv,
                     long kv_batch_stride, float* O, int ldo, long o_batch_stride, const int* q_pos,
                     const unsigned char* key_pad, int B, int Lq, int Lk, int A, hipStream_t st) {
  if (B <= 0 || Lq <= 0) return CTRLSIM_OK;
  if (Lk <= 0 || (ldq & 3) || (ldkv & 3)) return CTRLSIM_EINVAL;
  if (ctrlsim_option(OPT_ATTN_IMPL) == 1)
    return launch_attention_bf16x6(mode, Q, ldq, q_batch_stride, K, V, ldkv, kv_batch_stride, O, ldo, o_batch_stride, q_pos,
                                   key_pad, B, Lq, Lk, A, st);
  if (mode < 0 || mode > MODE_CAUSAL) return CTRLSIM_EINVAL;       // the IL / Trajeglish masks (modes 2, 3) exist in the bf16x6 kernel only
  dim3 g((Lq + 127) / 128, NHEAD, B), blk(256);
  const float scale = 0.17677669529663687f * 1.4426950408889634f;  // log2(e)/sqrt(32)
  // algorithmic FLOPs: (QK^T + PV) = 4*32 per visible (query, key) pair and head
  double pairs;
  if (mode == MODE_CAUSAL) {
    const double A3 = 3.0 * A;
    if (!q_pos) {   // all Lq = Lk tokens: per timestep block of A3 queries, keys of earlier steps + in-step pattern
      const double T = (double)Lq / A3;
      pairs = A3 * A3 * T * (T - 1) / 2.0 + T * A * (1.0 * (1 + A - 1) + (2 + A - 1) + (3 + A - 1));
    } else {        // gathered current-timestep tokens: bounded by the keys up to and including their timestep
      pairs = (double)Lq * (double)Lk;
    }
  } else {
    pairs = (double)Lq * (double)Lk;
  }
  prof_before(PROF_ATTN, st);
  if (mode == MODE_CAUSAL) {
    hipLaunchKernelGGL((attention_f32_kernel<MODE_CAUSAL>), g, blk, 0, st, Q, ldq, q_batch_stride, K, V, ldkv,
                       kv_batch_stride, O, ldo, o_batch_stride, q_pos, key_pad, Lq, Lk, A, scale);
  } else {
    if (!key_pad) return CTRLSIM_EINVAL;
    hipLaunchKernelGGL((attention_f32_kernel<MODE_KEYPAD>), g, blk, 0, st, Q, ldq, q_batch_stride, K, V, ldkv,
                       kv_batch_stride, O, ldo, o_batch_stride, q_pos, key_pad, Lq, Lk, A, scale);
  }
  prof_after(PROF_ATTN, pairs * 128.0 * NHEAD * B, st, (double)B * (8.0 * DM * Lq + 8.0 * DM * Lk));
  return ctrlsim_launch_status();
}
