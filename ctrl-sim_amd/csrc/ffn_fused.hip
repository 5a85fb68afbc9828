// Fused post-LN feed-forward block of nn.TransformerEncoderLayer / nn.TransformerDecoderLayer (norm_first = False):
//
//     y = LayerNorm( x + W2 . relu(W1 . x + b1) + b2 ) * gamma + beta          x, y: [M, 256] fp32 (y may alias x)
//
// as ONE kernel with fp32-class accuracy on the bf16 MFMA (operand split x = hi + mid + lo, six partial products per
// fp32 product: see gemm_bf16x6.hip).  The [M, F] hidden activation never exists in memory:
//
//   * one wave owns 32 rows of x for the whole block.  Their transposed, split fragments X^T (the MFMA B operand of the
//     first product: 16 k-steps x 3 planes = 192 registers) and the output accumulators Y^T [256 x 32] (128 registers)
//     stay in the register file — one wave per SIMD, the 512-register budget of a CDNA4 wave is what makes this fit.
//   * per block of 32 hidden units:  H^T = W1_blk . X^T  (96 MFMA, A = W1 fragments from LDS, accumulator initialised
//     with b1), ReLU + split in registers, and — exactly like P in the attention kernel — the accumulator-register
//     order IS the k-slot order of the second product  Y^T += W2_blk . H^T  (96 MFMA): H never moves between lanes.
//     W2's k-slots are permuted accordingly at pack time (ctrlsim_amd/pack.py:ffn_planes).
//   * the weight blocks (48 KB each: W1_blk, W2_blk alternating) stream through a 3-slot LDS ring by LDS-DMA, each
//     issued two phases ahead; the 4 waves of a workgroup (128 rows) share them.  One barrier per phase (96 MFMA).
//   * epilogue: Y^T through LDS (the ring is free by then) to row-major, + b2 + x, LayerNorm, 16-byte stores.
//
// HBM traffic: x read twice (operand + residual, the second an L2 hit) and y written once = 2-3 KB per row, against
// 13 KB per row for Linear / Linear+LN kernels with the hidden tensor in HBM.  Weights (3 MB of planes per block) come
// from L2.
#include "common.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ unsigned ffn_cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void ffn_split3_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = ffn_cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xFFFF0000u);
  mid = ffn_cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xFFFF0000u);
  lo = ffn_cvt_pk_bf16(sa, sb);
}
// eight fp32 values -> three bf16x8 fragments
__device__ __forceinline__ void ffn_split3_frag(const float* x, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  u32x4 h, m, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned a, b, c;
    ffn_split3_pair(x[2 * i], x[2 * i + 1], a, b, c);
    h[i] = a; m[i] = b; l[i] = c;
  }
  hi = __builtin_bit_cast(bf16x8, h);
  mid = __builtin_bit_cast(bf16x8, m);
  lo = __builtin_bit_cast(bf16x8, l);
}

constexpr int FF_BLK = 3 * 16 * 2 * 32 * 8;        // bf16 elements of one weight block (W1: [3][16][2][32][8]; W2: [3][2][2][256][8])
constexpr int FF_RING = 3;
#ifndef FFN_PF
#define FFN_PF 2                                   // LDS fragment prefetch distance in k-steps (2 or 3; four register buffers)
#endif
constexpr int FF_CP = DM + 4;                      // row pitch (floats) of the epilogue staging

// six partial products, smallest first
#define FFN_TERMS(ACC, A, B)                                                     \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], ACC, 0, 0, 0);       \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], ACC, 0, 0, 0);

// (Two independent accumulator chains per product were tried — a single dependent chain runs the matrix pipe at ~73 % with
// one wave per SIMD — but at this register pressure hipcc answers with v_accvgpr_mov shuffles / spills and the result is
// slower: profiles/r01_c_pmc_pipes.md.)
__global__ __launch_bounds__(256, 1) void ffn_fused_bf16x6_kernel(
    const float* X, int ldx, const __bf16* __restrict__ W1p, const float* __restrict__ b1,   // X may alias Y: no restrict
    const __bf16* __restrict__ W2p, const float* __restrict__ b2, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* Y, int ldy, int M, int nhb) {
  extern __shared__ __attribute__((aligned(16))) __bf16 ring[];      // FF_RING blocks of 48 KB, then b1 (F floats)
  float* b1s = reinterpret_cast<float*>(ring + FF_RING * FF_BLK);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int n_rb = (M + 127) / 128;
  for (int i = tid; i < nhb * 32; i += 256) b1s[i] = b1[i];          // global loads inside a phase would queue behind its DMA

  // block i of the weight stream: W1 of hidden block i/2 (even i) or W2 of it (odd i); lives in ring slot i % 3
  auto dma_block = [&](int i, int to_slot) {
    const __bf16* src = ((i & 1) ? W2p : W1p) + (size_t)(i >> 1) * FF_BLK + tid * 8;
    __bf16* dst = ring + to_slot * FF_BLK + wave * 64 * 8;            // wave-uniform LDS base (+ 16 B per lane)
#pragma unroll
    for (int j = 0; j < 12; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 256 * 8),
                                       (__attribute__((address_space(3))) void*)(dst + j * 256 * 8), 16, 0, 0);
  };
  // one 1 KB piece (per wave) of block i: issued between the MFMAs of a phase instead of as a burst of 12 at its start —
  // a piece costs the wave ~60-80 issue cycles, which then overlap the matrix pipe instead of idling it (ablation: the
  // burst cost 18 % of the kernel)
  auto dma_piece = [&](const __bf16* src, __bf16* dst, int j) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 256 * 8),
                                     (__attribute__((address_space(3))) void*)(dst + j * 256 * 8), 16, 0, 0);
  };
  const int nblk = 2 * nhb;
  // end of a phase: the block issued in the PREVIOUS phase must have landed (it is read next phase); the 12 pieces issued
  // in this phase may stay in flight (vmcnt counts in order).  LDS reads of this wave are complete (lgkmcnt(0)).
  auto phase_barrier = [&](bool issued_this_phase) {
#ifdef ABL_NO_BARRIER
    return;
#endif
    if (issued_this_phase) __builtin_amdgcn_s_waitcnt(0x0070 | 12);   // vmcnt(12) expcnt(7) lgkmcnt(0)  [gfx9: vmcnt = bits 3:0 + 15:14]
    else __builtin_amdgcn_s_waitcnt(0x0070);                          // nothing newer in flight: vmcnt(0)
    __builtin_amdgcn_s_barrier();
  };

  for (int rb = blockIdx.x; rb < n_rb; rb += gridDim.x) {
    // ---- this wave's 32 rows of x as split B-operand fragments: k-step ks covers k = 16 ks + 8 half .. + 7
    const int row = rb * 128 + wave * 32 + l31;
    const int rowc = row < M ? row : M - 1;                           // rows beyond M are computed on a clamped row, never stored
    dma_block(0, 0);
    dma_block(1, 1);
    bf16x8 xT[16][3];
    {
      const float* xp = X + (size_t)rowc * ldx + half * 8;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xp + ks * 16);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(xp + ks * 16 + 4);
        const float xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        ffn_split3_frag(xs, xT[ks][0], xT[ks][1], xT[ks][2]);
      }
    }
    f32x16 yacc[8];
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) yacc[ob][r] = 0.f;
    __syncthreads();                                                  // (vmcnt(0) + barrier) blocks 0 and 1 are in LDS

    int slot = 0;                                                     // ring slot of the block the current phase reads
    for (int hb = 0; hb < nhb; ++hb) {
      // ---------------- phase 2 hb: H^T = W1_blk . X^T (+ b1), block 2 hb in slot (2 hb) % 3
      // the block two ahead goes to the slot read last phase.  Past the end of the stream the last block is fetched again
      // (into a slot nobody reads any more; drained before the epilogue): the phases stay branch-free.
      const int hb_next = hb + 1 < nhb ? hb + 1 : nhb - 1;
      const __bf16* dsrc_a = W1p + (size_t)hb_next * FF_BLK + tid * 8;
      __bf16* ddst_a = ring + (slot == 0 ? 2 : slot - 1) * FF_BLK + wave * 64 * 8;
      f32x16 hacc;
      {
        const float* bp = b1s + hb * 32 + 4 * half;                    // register r <-> hidden (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + 8 * g);
          hacc[4 * g + 0] = bv[0]; hacc[4 * g + 1] = bv[1]; hacc[4 * g + 2] = bv[2]; hacc[4 * g + 3] = bv[3];
        }
      }
      {
        // fragments are fetched two k-steps ahead of the MFMAs that consume them (one wave per SIMD: nothing else hides
        // the LDS latency; hipcc does not hoist the reads by itself)
        const __bf16* w1 = ring + slot * FF_BLK + (half * 32 + l31) * 8;   // [p][ks][half][row][8]
        bf16x8 wf[4][3];
        auto ld1 = [&](int ks, bf16x8 (&f)[3]) {
#pragma unroll
#ifndef ABL_NO_FRAG
          for (int p = 0; p < 3; ++p) f[p] = *reinterpret_cast<const bf16x8*>(w1 + ((p * 16 + ks) * 2) * 32 * 8);
#else
          for (int p = 0; p < 3; ++p) { f[p] = xT[ks][p]; asm volatile("" : "+v"(f[p])); }
#endif
        };
        ld1(0, wf[0]);
        ld1(1, wf[1]);
        if (FFN_PF == 3) ld1(2, wf[2]);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (ks + FFN_PF < 16) ld1(ks + FFN_PF, wf[(ks + FFN_PF) & 3]);
#ifndef ABL_NO_DMA
          if (ks < 12) dma_piece(dsrc_a, ddst_a, ks);
#endif
          FFN_TERMS(hacc, wf[ks & 3], xT[ks])
        }
      }
      // ReLU + split: k-step kk of the second product uses accumulator registers 8 kk .. 8 kk + 7
      bf16x8 hf[2][3];
      {
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = fmaxf(hacc[r], 0.f);
#ifndef ABL_NO_SPLIT
        ffn_split3_frag(hv, hf[0][0], hf[0][1], hf[0][2]);
        ffn_split3_frag(hv + 8, hf[1][0], hf[1][1], hf[1][2]);
#else
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            u32x4 u = {__float_as_uint(hv[8 * kk + p]), __float_as_uint(hv[8 * kk + p + 1]), __float_as_uint(hv[8 * kk + p + 2]), __float_as_uint(hv[8 * kk + p + 3])};
            hf[kk][p] = __builtin_bit_cast(bf16x8, u);
          }
#endif
      }
      phase_barrier(true);                               // block 2 hb + 1 has landed; everyone is done with this slot
      slot = slot == FF_RING - 1 ? 0 : slot + 1;

      // ---------------- phase 2 hb + 1: Y^T += W2_blk . H^T, block 2 hb + 1 in slot (2 hb + 1) % 3
      const __bf16* dsrc_b = W2p + (size_t)hb_next * FF_BLK + tid * 8;
      __bf16* ddst_b = ring + (slot == 0 ? 2 : slot - 1) * FF_BLK + wave * 64 * 8;
      {
        const __bf16* w2 = ring + slot * FF_BLK + (half * 256 + l31) * 8;   // [p][kk][half][o][8]
        bf16x8 wf[4][3];
        auto ld2 = [&](int i, bf16x8 (&f)[3]) {        // step i = (out block ob = i >> 1, k-step kk = i & 1)
#pragma unroll
          for (int p = 0; p < 3; ++p)
#ifndef ABL_NO_FRAG
            f[p] = *reinterpret_cast<const bf16x8*>(w2 + (((p * 2 + (i & 1)) * 2) * 256 + (i >> 1) * 32) * 8);
#else
          { f[p] = hf[i & 1][p]; asm volatile("" : "+v"(f[p])); }
#endif
        };
        ld2(0, wf[0]);
        ld2(1, wf[1]);
        if (FFN_PF == 3) ld2(2, wf[2]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i + FFN_PF < 16) ld2(i + FFN_PF, wf[(i + FFN_PF) & 3]);
#ifndef ABL_NO_DMA
          if (i < 12) dma_piece(dsrc_b, ddst_b, i);
#endif
          FFN_TERMS(yacc[i >> 1], wf[i & 3], hf[i & 1])
        }
      }
      phase_barrier(true);
      slot = slot == FF_RING - 1 ? 0 : slot + 1;
    }
    __syncthreads();                                                  // drain everything before the ring is reused as staging

    // ---------------- epilogue: Y^T -> LDS (own 32-row region), then row-major + b2 + x, LayerNorm, store
    float* Cs = reinterpret_cast<float*>(ring) + wave * 32 * FF_CP;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) Cs[l31 * FF_CP + ob * 32 + mfma_row(r, half)] = yacc[ob][r];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                               // lgkmcnt(0): this wave's LDS writes are complete
    {
      const int col = lane * 4;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + col);
      const f32x4 gg = *reinterpret_cast<const f32x4*>(gamma + col);
      const f32x4 be = *reinterpret_cast<const f32x4*>(beta + col);
#pragma unroll 4
      for (int rr = 0; rr < 32; ++rr) {
        const int grow = rb * 128 + wave * 32 + rr;
        if (grow >= M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(Cs + rr * FF_CP + col);
        v += bb;
        v += *reinterpret_cast<const f32x4*>(X + (size_t)grow * ldx + col);
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
        const f32x4 dv = v - mean;
        const float var = wave_sum(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2] + dv[3] * dv[3]) * (1.f / 256.f);
        const f32x4 y = dv * (1.0f / sqrtf(var + 1e-5f)) * gg + be;
        *reinterpret_cast<f32x4*>(Y + (size_t)grow * ldy + col) = y;
      }
    }
    __syncthreads();                                                  // the ring is reused by the next row block
  }
}

}  // namespace

// y = LayerNorm(x + W2 relu(W1 x + b1) + b2) * gamma + beta for rows of 256; W1p / W2p = pack.py:ffn_planes images of
// F = 32 * nhb hidden units.  y may alias x (a row is read completely before it is written, by the same wave).
int launch_ffn_fused_bf16x6(const float* X, int ldx, const void* W1p, const float* b1, const void* W2p, const float* b2,
                            const float* gamma, const float* beta, float* Y, int ldy, int M, int F, hipStream_t st) {
  if (M <= 0) return CTRLSIM_OK;
  if (!X || !W1p || !W2p || !b1 || !b2 || !gamma || !beta || !Y || (ldx & 3) || (ldy & 3) || F <= 0 || (F & 31) || F > 4096)
    return CTRLSIM_EINVAL;
  const int n_rb = (M + 127) / 128;
  const int grid = n_rb < 256 ? n_rb : 256;                           // one persistent workgroup per CU
  const size_t shm = (size_t)FF_RING * FF_BLK * sizeof(__bf16) + (size_t)F * sizeof(float);   // 147456 B + b1
  // once per process (thread-safe static initialisation), sized for the largest F this launcher accepts
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_bf16x6_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)(FF_RING * FF_BLK * sizeof(__bf16) + 4096 * sizeof(float))) == hipSuccess;
  if (!attr_ok) return CTRLSIM_EINVAL;
  prof_before(PROF_GEMM, st);
  hipLaunchKernelGGL(ffn_fused_bf16x6_kernel, dim3(grid), dim3(256), shm, st, X, ldx, static_cast<const __bf16*>(W1p), b1,
                     static_cast<const __bf16*>(W2p), b2, gamma, beta, Y, ldy, M, F / 32);
  prof_after(PROF_GEMM, 4.0 * (double)M * DM * (double)F, st, 12.0 * (double)M * DM + 12.0 * (double)DM * F);
  return ctrlsim_launch_status();
}
