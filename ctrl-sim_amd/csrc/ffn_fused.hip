// Fused post-LN feed-forward block of nn.TransformerEncoderLayer / nn.TransformerDecoderLayer (norm_first = False):
//
//     y = LayerNorm( x + W2 . relu(W1 . x + b1) + b2 ) * gamma + beta          x, y: [M, 256] fp32 (y may alias x)
//
// as ONE kernel with fp32-class accuracy on the 16-bit MFMA (operand split into NPL planes, csrc/split.h: two fp16 planes and
// three partial products per fp32 product by default; register / block sizes below are quoted for three bf16 planes).  The [M, F] hidden activation never exists in memory:
//
//   * one wave owns 32 rows of x for the whole block.  Their transposed, split fragments X^T (the MFMA B operand of the
//     first product: 16 k-steps x 3 planes = 192 registers) and the output accumulators Y^T [256 x 32] (128 registers)
//     stay in the register file — one wave per SIMD, the 512-register budget of a CDNA4 wave is what makes this fit.
//   * per block of 32 hidden units:  H^T = W1_blk . X^T  (96 MFMA, A = W1 fragments from LDS, accumulator initialised
//     with b1), ReLU + split in registers, and — exactly like P in the attention kernel — the accumulator-register
//     order IS the k-slot order of the second product  Y^T += W2_blk . H^T  (96 MFMA): H never moves between lanes.
//     W2's k-slots are permuted accordingly at pack time (ctrlsim_amd/pack.py:ffn_planes).
//   * the weight blocks (48 KB each: W1_blk, W2_blk alternating) stream through an LDS ring by LDS-DMA (four 32 KB slots
//     and one barrier per hidden block with two fp16 planes, round 4: +1.1 % on the kernel; three 48 KB slots with three bf16 planes), each
//     issued two phases ahead; the 4 waves of a workgroup (128 rows) share them.  One barrier per phase (96 MFMA); per pair of phases with four slots.
//   * epilogue: Y^T through LDS (the ring is free by then) to row-major, + b2 + x, LayerNorm, 16-byte stores.
//
// HBM traffic: x read twice (operand + residual) and y written once = 3 KB per row, against
// 13 KB per row for Linear / Linear+LN kernels with the hidden tensor in HBM.  Weights (3 MB of planes per block) come
// from L2.
//
// PRE (round 5, two-plane scheme): the post-LN block in FRONT of the feed-forward block — x <- LayerNorm(x + Wo . o + bo), the attention
// out-projection with its residual and LayerNorm (decoder: multihead_attn.out_proj + norm2, encoder: self_attn.out_proj + norm1;
// nn.TransformerDecoderLayer / EncoderLayer, modules/decoder.py:16-20, encoder.py:42-46) — runs as a LEADING product of the same kernel:
// the wave's 32 rows of the attention output o are the first B operand, Wo streams through the ring as eight W1-shaped blocks into the Y
// accumulators (8 x 48 MFMA), the LayerNorm runs in the accumulator layout (a lane holds 128 values of its row, the other 128 sit in lane ^ 32:
// one v_permlane32_swap per statistic), its output is split IN PLACE into the X^T fragments — the accumulator-register order is the k-slot
// order of a W1 image with bits 2 and 3 of k swapped (pack.py: ffn_planes_pre) — and stays in the accumulators as the residual of the
// feed-forward block, so x is neither written nor re-read in between: 3 KB per row for both blocks instead of 6.
#include "split.h"
#include <type_traits>

namespace SPLIT_NS {

namespace {

constexpr int FF_BLK = NPL * 16 * 2 * 32 * 8;      // 16-bit elements of one weight block (W1: [NPL][16][2][32][8]; W2: [NPL][2][2][256][8])
constexpr int FF_PIECES = FF_BLK / (256 * 8);      // 16-byte-per-thread LDS-DMA pieces of a block (8 / 12)
// two planes: four ring slots (128 KB) — both blocks of the NEXT hidden block are requested while the current one is computed, and the
// workgroup meets once per hidden block (after the second product) instead of once per product; three planes: three 48 KB slots, a
// meeting per product
constexpr bool FF_PAIR = NPL == 2;
constexpr int FF_RING = FF_PAIR ? 4 : 3;
constexpr int FFN_PF = NPL == 2 ? 3 : 2;           // LDS fragment prefetch distance in k-steps (four register buffers); 3 needs the
                                                   // registers the two-plane scheme frees (+1.5 %), with three planes it spilled
constexpr int FF_CP = DM + 4;                      // row pitch (floats) of the epilogue staging
// LDS: the ring, re-used by the epilogue as 4 x 32 rows of FF_CP floats — whichever is larger — then b1 (F floats)
constexpr size_t FF_RING_BYTES = (size_t)FF_RING * FF_BLK * sizeof(op_t) > (size_t)4 * 32 * FF_CP * 4
                                     ? (size_t)FF_RING * FF_BLK * sizeof(op_t) : (size_t)4 * 32 * FF_CP * 4;

#define FFN_TERMS(ACC, A, B) SPLIT_TERMS(ACC, A, B)
#ifndef FF_2CH
#define FF_2CH 0          // 1: the W1-shaped products on two alternating accumulator chains (round 6 experiment)
#endif

// (Two independent accumulator chains per product were tried — a single dependent chain runs the matrix pipe at ~73 % with
// one wave per SIMD — but at this register pressure hipcc answers with v_accvgpr_mov shuffles / spills and the result is
// slower: profiles/r01_c_pmc_pipes.md.)
struct FfnPre {                      // PRE: residual rows, Wo blocks, bo, norm gain / bias; QP: where the LayerNorm output rows go
  const float* R; int ldr; const op_t* Wop; const float* bo; const float* g0; const float* be0; float* X1; int ldx1;
};
// MODE 0 = the feed-forward block; 1 = PRE: out-projection + residual + LayerNorm in front of it; 2 = QP: that leading block followed by ONE
// 256 -> 256 Linear instead of the feed-forward block (the cross-attention query projection: x1 and q leave, W1p = the eight blocks of Wq)
template <int MODE>
__global__ __launch_bounds__(256, 1) void ffn_fused_bf16x6_kernel(
    const float* X, int ldx, FfnPre pa, const op_t* __restrict__ W1p, const float* __restrict__ b1,   // X may alias Y: no restrict (PRE: X = the attention output)
    const op_t* __restrict__ W2p, const float* __restrict__ b2, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* Y, int ldy, int M, int nhb, int* __restrict__ nonfinite) {
  constexpr bool PRE = MODE != 0, QP = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) op_t ring[];      // FF_RING blocks of 48 KB, then b1 (F floats)
  float* b1s = reinterpret_cast<float*>(reinterpret_cast<char*>(ring) + FF_RING_BYTES);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int n_rb = (M + 127) / 128;
  // PRE / QP: the wave index as a scalar — the LDS-DMA destinations of the unrolled leading products (slot and piece are compile-time
  // constants there) are then scalar arithmetic instead of 32 loop-invariant vector registers
  const int wave_d = PRE ? __builtin_amdgcn_readfirstlane(wave) : wave;
  for (int i = tid; i < nhb * 32; i += 256) b1s[i] = b1[i];          // global loads inside a phase would queue behind its DMA
  static_assert(!PRE || FF_PAIR, "the leading product rides on the four-slot ring of the two-plane scheme");
  float* pre_s = b1s + nhb * 32;                                     // PRE: bo, norm gain, norm bias, b2 (256 floats each)
  if (PRE) { pre_s[tid] = pa.bo[tid]; pre_s[256 + tid] = pa.g0[tid]; pre_s[512 + tid] = pa.be0[tid]; pre_s[768 + tid] = b2[tid]; }
  if (PRE && !QP) { pre_s[1024 + tid] = gamma[tid]; pre_s[1280 + tid] = beta[tid]; }      // the closing LayerNorm's gain / bias
  __syncthreads();

  // block i of the weight stream: W1 of hidden block i/2 (even i) or W2 of it (odd i); lives in ring slot i % FF_RING
  auto dma_block = [&](int i, int to_slot) {
    const op_t* src = (PRE ? pa.Wop : (i & 1) ? W2p : W1p) + tid * 8;   // (called for blocks 0 and 1 only)
    if (PRE) asm volatile("" : "+v"(src));                            // (PRE: or the sixteen piece addresses are kept across the row-block loop — in scratch)
    src += (size_t)(PRE ? i : (i >> 1)) * FF_BLK;
    op_t* dst = ring + to_slot * FF_BLK + wave_d * 64 * 8;            // wave-uniform LDS base (+ 16 B per lane)
#pragma unroll
    for (int j = 0; j < FF_PIECES; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 256 * 8),
                                       (__attribute__((address_space(3))) void*)(dst + j * 256 * 8), 16, 0, 0);
  };
  // one 1 KB piece (per wave) of block i: issued between the MFMAs of a phase instead of as a burst of 12 at its start —
  // a piece costs the wave ~60-80 issue cycles, which then overlap the matrix pipe instead of idling it (ablation: the
  // burst cost 18 % of the kernel)
  auto dma_piece = [&](const op_t* src, op_t* dst, int j) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 256 * 8),
                                     (__attribute__((address_space(3))) void*)(dst + j * 256 * 8), 16, 0, 0);
  };
  const int nblk = 2 * nhb;
  // end of a phase: the block issued in the PREVIOUS phase must have landed (it is read next phase); the 12 pieces issued
  // in this phase may stay in flight (vmcnt counts in order).  LDS reads of this wave are complete (lgkmcnt(0)).
  auto phase_barrier = [&](bool issued_this_phase) {
    if (issued_this_phase) __builtin_amdgcn_s_waitcnt(0x0070 | FF_PIECES);   // vmcnt(pieces) expcnt(7) lgkmcnt(0)  [gfx9: vmcnt = bits 3:0 + 15:14]
    else __builtin_amdgcn_s_waitcnt(0x0070);                          // nothing newer in flight: vmcnt(0)
    __builtin_amdgcn_s_barrier();
  };

  // PRE / QP: all 32 loads of the wave's rows are requested before the first is converted (left to itself hipcc, at this kernel's register
  // pressure, requests two, waits for them, converts, requests the next two: sixteen round trips to HBM per row block).  MODE 1 requests the
  // NEXT row block's rows in front of its epilogue — the fragment registers are dead there — so that they arrive underneath the closing
  // LayerNorm and the stores (s_memtime: the wait for the rows was 6.4 % of the mode)
  constexpr bool ROWS_AHEAD = MODE == 1;
  f32x4 raw[PRE ? 32 : 1];
  // (the lane index goes through an opaque point in these helpers: hipcc otherwise keeps the loop-invariant 64-bit part of every row address
  // across the row-block loop — in scratch at this register pressure, and a scratch reload in front of a batch of loads is a vmcnt(0))
  auto issue_rows = [&](int rb_) {
    int ln_ = lane;
    if (PRE) asm volatile("" : "+v"(ln_));
    const int r_ = rb_ * 128 + wave * 32 + (ln_ & 31);
    const float* xp = X + ((size_t)(r_ < M ? r_ : M - 1) * ldx + (ln_ >> 5) * 8);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      raw[2 * ks] = *reinterpret_cast<const f32x4*>(xp + ks * 16);
      raw[2 * ks + 1] = *reinterpret_cast<const f32x4*>(xp + ks * 16 + 4);
    }
  };
  f32x4 rr[PRE ? 32 : 1];                                             // the residual rows, in the accumulator layout (below)
  auto issue_residual = [&](int rb_) {
    int ln_ = lane;
    asm volatile("" : "+v"(ln_));
    const int r_ = rb_ * 128 + wave * 32 + (ln_ & 31);
    const float* rp = pa.R + ((size_t)(r_ < M ? r_ : M - 1) * pa.ldr + 4 * (ln_ >> 5));
#pragma unroll
    for (int i = 0; i < 32; ++i) rr[i] = *reinterpret_cast<const f32x4*>(rp + (i >> 2) * 32 + 8 * (i & 3));
  };
  if constexpr (ROWS_AHEAD) { issue_rows(blockIdx.x); issue_residual(blockIdx.x); }
  for (int rb = blockIdx.x; rb < n_rb; rb += gridDim.x) {
    // ---- this wave's 32 rows of x as split B-operand fragments: k-step ks covers k = 16 ks + 8 half .. + 7
    const int row = rb * 128 + wave * 32 + l31;
    const int rowc = row < M ? row : M - 1;                           // rows beyond M are computed on a clamped row, never stored
    dma_block(0, 0);
    dma_block(1, 1);
    opx8 xT[16][NPL];
    if constexpr (PRE) {
      if constexpr (!ROWS_AHEAD) issue_rows(rb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const f32x4 x0 = raw[2 * ks], x1 = raw[2 * ks + 1];
        const float xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        split_frag(xs, xT[ks]);
      }
    } else {
      const float* xp = X + (size_t)rowc * ldx + half * 8;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xp + ks * 16);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(xp + ks * 16 + 4);
        const float xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        split_frag(xs, xT[ks]);
      }
    }
    __syncthreads();                                                  // (vmcnt(0) + barrier) blocks 0 and 1 are in LDS
    f32x16 yacc[8];
    if (PRE) {
      // the residual rows + bo go straight into the accumulators of the leading product, in the accumulator layout — lane (l31, half) holds, of
      // row l31, the columns 32 ob + (r & 3) + 8 (r >> 2) + 4 half — and in the scale of the Wo planes.  Requested BEHIND the barrier: out-block
      // ob is first needed by product ob, so all but the first of these loads land underneath the leading product's MFMAs
      if constexpr (!ROWS_AHEAD) issue_residual(rb);  // (MODE 1: requested in front of the previous row block's epilogue, like the rows above)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 xr = rr[4 * ob + g];
          const f32x4 bb = *reinterpret_cast<const f32x4*>(pre_s + ob * 32 + 8 * g + 4 * half);
#pragma unroll
          for (int e = 0; e < 4; ++e) yacc[ob][4 * g + e] = (xr[e] + bb[e]) * WSCALE;
        }
    } else {
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[ob][r] = 0.f;
    }

    int slot = 0;                                                     // ring slot of the block the current phase reads
    // one product of the W1 shape: acc^T += Wblk . X^T with the 32 KB block in ring slot `slot_` (A fragments from LDS, fetched FFN_PF
    // k-steps ahead of the MFMAs that consume them: one wave per SIMD, nothing else hides the LDS latency and hipcc does not hoist the
    // reads by itself), B = the wave's X^T fragments; the pieces of a later block are requested between the MFMAs
    auto product_w1 = [&](f32x16& acc, int slot_, const op_t* dsrc, op_t* ddst, bool dma = true, size_t dsrc_off = 0) {
      // (opaque to the optimiser: the unrolled leading products otherwise get their 64-bit per-lane piece addresses hoisted out of the
      // row-block loop — 128 registers of loop-invariant pointers, i.e. spills; the block offset is added BEHIND the opaque point, so
      // that one per-lane base is all that can be kept across iterations)
      asm volatile("" : "+v"(dsrc));
      dsrc += dsrc_off;
      const op_t* w1 = ring + slot_ * FF_BLK + (half * 32 + l31) * 8;   // [p][ks][half][row][8]
      opx8 wf[4][NPL];
      auto ld1 = [&](int ks, opx8 (&f)[NPL]) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) f[p] = *reinterpret_cast<const opx8*>(w1 + ((p * 16 + ks) * 2) * 32 * 8);
      };
      ld1(0, wf[0]);
      ld1(1, wf[1]);
      if (FFN_PF == 3) ld1(2, wf[2]);
#if FF_2CH
      // two accumulator chains (even / odd k-steps), their MFMAs alternating: every MFMA then follows one on the OTHER accumulator.  The 48
      // products of a block are otherwise ONE dependent chain with two LDS reads, a DMA piece and their waits between its links — and an
      // issue slot between two MFMAs on the same accumulator costs ~43 cycles, between different accumulators ~6 (MI355X_MICROARCH.md)
      f32x16 accb;
#pragma unroll
      for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 16; ks += 2) {
        // (the four fragment buffers hold the pair in flight and the next pair: requested one pair = six MFMAs ahead)
        if (ks + 2 < 16 && !(FFN_PF == 3 && ks == 0)) ld1(ks + 2, wf[(ks + 2) & 3]);
        if (ks + 3 < 16) ld1(ks + 3, wf[(ks + 3) & 3]);
        if (dma && ks < FF_PIECES) dma_piece(dsrc, ddst, ks);
        if (dma && ks + 1 < FF_PIECES) dma_piece(dsrc, ddst, ks + 1);
#define FF_ALT(PA, PB)                                                   \
  acc = MFMA_OP(wf[ks & 3][PA], xT[ks][PB], acc);                       \
  accb = MFMA_OP(wf[(ks + 1) & 3][PA], xT[ks + 1][PB], accb);
        PROD_LIST(FF_ALT)
#undef FF_ALT
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += accb[r];
#else
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks + FFN_PF < 16) ld1(ks + FFN_PF, wf[(ks + FFN_PF) & 3]);
        if (dma && ks < FF_PIECES) dma_piece(dsrc, ddst, ks);
        FFN_TERMS(acc, wf[ks & 3], xT[ks])
      }
#endif
    };
    if (PRE) {
      // ---------------- leading product: Y^T = Wo . O^T, eight W1-shaped blocks (two per barrier, like a hidden block's pair); the pair
      // after the last one is the first hidden block's (W1, W2)
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const op_t* dsrc = (pi < 3 ? pa.Wop : (QP || !s2) ? W1p : W2p) + tid * 8;
          op_t* ddst = ring + ((slot + 2) & 3) * FF_BLK + wave_d * 64 * 8;
          product_w1(yacc[2 * pi + s2], slot, dsrc, ddst, true, (size_t)(pi < 3 ? 2 * (pi + 1) + s2 : QP ? s2 : 0) * FF_BLK);
          if (s2) phase_barrier(false);
          slot = (slot + 1) & 3;
        }
      }
      // ---------------- x <- LayerNorm(x + Wo o + bo) in the accumulator layout; the other half of a lane's row lives in lane ^ 32
      float sum = 0.f;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = yacc[ob][r] * WSCALE_INV;
          yacc[ob][r] = v;
          sum += v;
        }
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
        sum = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      const float mean = sum * (1.f / 256.f);
      float sq = 0.f;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float dv = yacc[ob][r] - mean; sq = fmaf(dv, dv, sq); }
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
        sq = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      const float var = sq * (1.f / 256.f);
      if (half == 0 && row < M && !(var <= 3.0e38f)) atomicAdd(nonfinite, 1);      // one count per row whose variance is not finite
      const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = ob * 32 + 8 * g + 4 * half;
          const f32x4 gg = *reinterpret_cast<const f32x4*>(pre_s + 256 + c0);
          const f32x4 be = *reinterpret_cast<const f32x4*>(pre_s + 512 + c0);
#pragma unroll
          for (int e = 0; e < 4; ++e) yacc[ob][4 * g + e] = fmaf((yacc[ob][4 * g + e] - mean) * rstd, gg[e], be[e]);
        }
        // the LayerNorm output as the feed-forward block's B operand: k-step 2 ob + kk <- registers 8 kk .. 8 kk + 7 (W1 image with
        // the matching k order), and — plus b2, in the scale of the W2 planes — as the initial value of its Y accumulators (the residual)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          float t8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) t8[j] = yacc[ob][8 * kk + j];
          split_frag(t8, xT[2 * ob + kk]);
        }
        if (!QP) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 b2v = *reinterpret_cast<const f32x4*>(pre_s + 768 + ob * 32 + 8 * g + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) yacc[ob][4 * g + e] = (yacc[ob][4 * g + e] + b2v[e]) * WSCALE;
          }
        }
        __builtin_amdgcn_sched_barrier(0);               // one out-block at a time: hipcc otherwise hoists all 96 LDS reads (spills)
      }
    }
    if constexpr (QP) {
      // ---------------- Q^T = Wq . X1^T + bq: eight more W1-shaped blocks, two per barrier; blocks 0 and 1 were requested under the last pair above.
      // Both outputs leave from the ACCUMULATOR layout (a lane pair writes 32 contiguous bytes of its row; the four stores of an out-block
      // complete its 128-byte line in L2), spread over the products so that they drain underneath the MFMAs: block ob of x1 right before its
      // registers become the accumulator of q block ob, q block ob two products later.  X1 may alias R: a lane overwrites what it read.
      auto store_q = [&](int ob) {
        if (row < M) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(Y + (size_t)row * ldy + ob * 32 + 8 * g + 4 * half) =
                f32x4{yacc[ob][4 * g], yacc[ob][4 * g + 1], yacc[ob][4 * g + 2], yacc[ob][4 * g + 3]} * WSCALE_INV;
        }
      };
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int ob = 2 * pi + s2;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c0 = ob * 32 + 8 * g + 4 * half;
            if (row < M)
              *reinterpret_cast<f32x4*>(pa.X1 + (size_t)row * pa.ldx1 + c0) =
                  f32x4{yacc[ob][4 * g], yacc[ob][4 * g + 1], yacc[ob][4 * g + 2], yacc[ob][4 * g + 3]};
            const f32x4 bqv = *reinterpret_cast<const f32x4*>(pre_s + 768 + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) yacc[ob][4 * g + e] = bqv[e] * WSCALE;
          }
          if (ob >= 2) store_q(ob - 2);
          op_t* ddst = ring + ((slot + 2) & 3) * FF_BLK + wave_d * 64 * 8;
          product_w1(yacc[2 * pi + s2], slot, W1p + tid * 8, ddst, pi < 3, (size_t)(pi < 3 ? 2 * (pi + 1) + s2 : 0) * FF_BLK);
          if (s2) phase_barrier(false);
          slot = (slot + 1) & 3;
        }
      }
      store_q(6);
      store_q(7);
      // (every wave is past the last pair's barrier: nobody reads the ring any more and no request is in flight — the next row block may start)
    } else
    for (int hb = 0; hb < nhb; ++hb) {
      // ---------------- phase 2 hb: H^T = W1_blk . X^T (+ b1), block 2 hb in slot (2 hb) % FF_RING
      // the block two ahead goes to the slot read last phase.  Past the end of the stream the last block is fetched again
      // (into a slot nobody reads any more; drained before the epilogue): the phases stay branch-free.
      const int hb_next = hb + 1 < nhb ? hb + 1 : nhb - 1;
      const op_t* dsrc_a = W1p + (size_t)hb_next * FF_BLK + tid * 8;
      // pair barrier: block 2 hb + 2 -> the slot block 2 hb - 2 left a pair ago; otherwise the slot read last phase
      op_t* ddst_a = ring + (FF_PAIR ? ((slot + 2) & 3) : (slot == 0 ? 2 : slot - 1)) * FF_BLK + wave_d * 64 * 8;
      f32x16 hacc;
      {
        const float* bp = b1s + hb * 32 + 4 * half;                    // register r <-> hidden (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + 8 * g);
          hacc[4 * g + 0] = bv[0] * WSCALE; hacc[4 * g + 1] = bv[1] * WSCALE; hacc[4 * g + 2] = bv[2] * WSCALE; hacc[4 * g + 3] = bv[3] * WSCALE;   // the W1 planes carry WSCALE
        }
      }
      product_w1(hacc, slot, dsrc_a, ddst_a);
      // ReLU + split: k-step kk of the second product uses accumulator registers 8 kk .. 8 kk + 7
      opx8 hf[2][NPL];
      {
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = fmaxf(hacc[r] * WSCALE_INV, 0.f);
        split_frag(hv, hf[0]);
        split_frag(hv + 8, hf[1]);
      }
      // (pair barrier: no meeting here — block 2 hb + 1 landed before this pair began and nothing overwrites a slot inside a pair)
      if (!FF_PAIR) phase_barrier(true);                 // block 2 hb + 1 has landed; everyone is done with this slot
      slot = slot == FF_RING - 1 ? 0 : slot + 1;

      // ---------------- phase 2 hb + 1: Y^T += W2_blk . H^T, block 2 hb + 1 in slot (2 hb + 1) % FF_RING
      const op_t* dsrc_b = W2p + (size_t)hb_next * FF_BLK + tid * 8;
      op_t* ddst_b = ring + (FF_PAIR ? ((slot + 2) & 3) : (slot == 0 ? 2 : slot - 1)) * FF_BLK + wave_d * 64 * 8;   // block 2 hb + 3 -> the slot of block 2 hb - 1
      {
        const op_t* w2 = ring + slot * FF_BLK + (half * 256 + l31) * 8;   // [p][kk][half][o][8]
        opx8 wf[4][NPL];
        // step i = (out block ob = i & 7, k-step kk = i >> 3): k-step-major — consecutive groups of three products go to DIFFERENT output
        // accumulators (round 5: +1 % over out-block-major, where six dependent products queue on one accumulator)
        auto ld2 = [&](int i, opx8 (&f)[NPL]) {
#pragma unroll
          for (int p = 0; p < NPL; ++p)
            f[p] = *reinterpret_cast<const opx8*>(w2 + (((p * 2 + (i >> 3)) * 2) * 256 + (i & 7) * 32) * 8);
        };
        ld2(0, wf[0]);
        ld2(1, wf[1]);
        if (FFN_PF == 3) ld2(2, wf[2]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i + FFN_PF < 16) ld2(i + FFN_PF, wf[(i + FFN_PF) & 3]);
          if (i < FF_PIECES) dma_piece(dsrc_b, ddst_b, i);
          FFN_TERMS(yacc[(i & 7)], wf[i & 3], hf[(i >> 3)])
        }
      }
      phase_barrier(!FF_PAIR);                           // pair barrier: vmcnt(0) — the next hidden block's two weight blocks have landed for every wave
      slot = slot == FF_RING - 1 ? 0 : slot + 1;
    }
    if constexpr (PRE && !QP) {
      // ---------------- epilogue of the PRE mode: the closing LayerNorm IN the accumulator layout, like the one in front (b2 and the residual are
      // in the accumulators; a lane holds 128 of its row's 256 values, the other half is in lane ^ 32), rows leave as 16-byte stores (a lane pair
      // writes 32 contiguous bytes, the four stores of an out-block complete its 128-byte line in L2).  No LDS staging, no barrier: every wave is
      // past the last pair's barrier, nobody reads the ring any more and no request is in flight.  (s_memtime: the LDS-staged row-major epilogue
      // below was 10.5 % of this mode, 23 000 ticks per row block against 8 000 for the in-register LayerNorm.)  Y may alias R or X: a wave
      // read all of its rows long ago.
      issue_rows(rb + gridDim.x);                      // the next row block's rows (past the last one: a clamped row, never used)
      issue_residual(rb + gridDim.x);                  // (rows of the NEXT row block: only this workgroup ever writes them, later)
      // (Round 6: guarding the two requests with `rb + gridDim.x < n_rb` — so that nothing is requested past the last row block — costs the
      // mode 644 bytes of scratch per lane at this register pressure (A 232 -> 256 + spills): not done.  Past the last block the lanes read
      // row M - 1 on a clamped address, possibly while its owner stores it; the values are never used: see the ALIASING RULE at the launcher.)
      __builtin_amdgcn_sched_barrier(0);
      float sum = 0.f;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = yacc[ob][r] * WSCALE_INV;
          yacc[ob][r] = v;
          sum += v;
        }
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
        sum = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      const float mean = sum * (1.f / 256.f);
      float sq = 0.f;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float dv = yacc[ob][r] - mean; sq = fmaf(dv, dv, sq); }
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
        sq = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      const float var = sq * (1.f / 256.f);
      if (half == 0 && row < M && !(var <= 3.0e38f)) atomicAdd(nonfinite, 1);      // one count per row whose variance is not finite
      const float rstd = 1.0f / sqrtf(var + 1e-5f);
      int row_o = row;
      asm volatile("" : "+v"(row_o));
      float* const yrow = Y + (size_t)row_o * ldy;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = ob * 32 + 8 * g + 4 * half;
          const f32x4 gg = *reinterpret_cast<const f32x4*>(pre_s + 1024 + c0);
          const f32x4 be = *reinterpret_cast<const f32x4*>(pre_s + 1280 + c0);
          f32x4 y;
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = fmaf((yacc[ob][4 * g + e] - mean) * rstd, gg[e], be[e]);
          if (row < M) *reinterpret_cast<f32x4*>(yrow + c0) = y;
        }
      }
    }
    if constexpr (!PRE) {
    __syncthreads();                                                  // drain everything before the ring is reused as staging

    // ---------------- epilogue: Y^T -> LDS (own 32-row region), then row-major + b2 + x, LayerNorm, store
    float* Cs = reinterpret_cast<float*>(ring) + wave * 32 * FF_CP;
    // the residual rows of this wave are requested before the accumulators go through LDS (the X^T fragment registers are dead
    // by now): their latency runs under the 128 ds_writes instead of in front of every row's reductions.  X may alias Y — all
    // of the wave's reads are issued before its first store.
    f32x4 xpre[32];       // all 32 residual rows of the wave, requested before the accumulators go through LDS (-4 %)
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
      const int grow = rb * 128 + wave * 32 + rr;
      xpre[rr] = grow < M ? *reinterpret_cast<const f32x4*>(X + (size_t)grow * ldx + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int r = 0; r < 16; ++r) Cs[l31 * FF_CP + ob * 32 + mfma_row(r, half)] = yacc[ob][r] * WSCALE_INV;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                               // lgkmcnt(0): this wave's LDS writes are complete
    {
      const int col = lane * 4;
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + col);
      const f32x4 gg = *reinterpret_cast<const f32x4*>(gamma + col);
      const f32x4 be = *reinterpret_cast<const f32x4*>(beta + col);
#pragma unroll
      for (int rr = 0; rr < 32; ++rr) {
        const int grow = rb * 128 + wave * 32 + rr;
        if (grow >= M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(Cs + rr * FF_CP + col);
        v += bb; v += xpre[rr];
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
        const f32x4 dv = v - mean;
        const float var = wave_sum(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2] + dv[3] * dv[3]) * (1.f / 256.f);
        count_nonfinite_row(var, lane, nonfinite);
        const f32x4 y = dv * (1.0f / sqrtf(var + 1e-5f)) * gg + be;
        *reinterpret_cast<f32x4*>(Y + (size_t)grow * ldy + col) = y;
      }
    }
    __syncthreads();                                                  // the ring is reused by the next row block
    }
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
  const size_t shm = FF_RING_BYTES + (size_t)F * sizeof(float);
  // once per process (thread-safe static initialisation), sized for the largest F this launcher accepts
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_bf16x6_kernel<0>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)(FF_RING_BYTES + 4096 * sizeof(float))) == hipSuccess;
  if (!attr_ok) return CTRLSIM_EINVAL;
  prof_before(PROF_GEMM, st);
  hipLaunchKernelGGL(ffn_fused_bf16x6_kernel<0>, dim3(grid), dim3(256), shm, st, X, ldx, FfnPre{}, static_cast<const op_t*>(W1p), b1,
                     static_cast<const op_t*>(W2p), b2, gamma, beta, Y, ldy, M, F / 32, ctrlsim_nonfinite_ptr());
  prof_after(PROF_GEMM, 4.0 * (double)M * DM * (double)F, st, 12.0 * (double)M * DM + 4.0 * NPL * (double)DM * F, PKIND_FFN);   // x read as operand and as residual, y written; W1 / W2 as NPL planes
  return ctrlsim_launch_status();
}

// y = LayerNorm3(x1 + W2 relu(W1 x1 + b1) + b2) with x1 = LayerNorm0(R + Wo O + bo): the attention out-projection + residual + LayerNorm
// and the feed-forward block behind it as ONE kernel (PRE above).  O = the attention output rows, R = the residual rows (y may alias R or O);
// ALIASING RULE: a wave reads all of its 32 rows (O and R) before it stores any, and only the wave that owns a row reads or writes it —
// with one exception that callers may rely on being harmless: lanes whose row is >= M (the tail of the LAST row block, and MODE 1's request
// for the row block past a workgroup's last one) read row M - 1 instead (clamped address), possibly while its owner stores it; what they
// read only feeds accumulator columns that are never stored or counted.
// Wop = pack.py:row_blocks(Wo) (eight 32-column blocks), W1q / W2p = pack.py:ffn_planes_pre.  Two-plane scheme only (CTRLSIM_EINVAL otherwise).
int launch_ffn_fused_pre(const float* O, int ldo, const float* R, int ldr, const void* Wop, const float* bo, const float* g0,
                         const float* be0, const void* W1q, const float* b1, const void* W2p, const float* b2, const float* gamma,
                         const float* beta, float* Y, int ldy, int M, int F, hipStream_t st) {
  if (M <= 0) return CTRLSIM_OK;
  if (!O || !R || !Wop || !bo || !g0 || !be0 || !W1q || !W2p || !b1 || !b2 || !gamma || !beta || !Y || (ldo & 3) || (ldr & 3) || (ldy & 3) ||
      F <= 0 || (F & 31) || F > 3072)
    return CTRLSIM_EINVAL;
  if constexpr (!FF_PAIR) {
    return CTRLSIM_EINVAL;
  } else {
    const int n_rb = (M + 127) / 128;
    const int grid = n_rb < 256 ? n_rb : 256;
    const size_t shm = FF_RING_BYTES + (size_t)(F + 1536) * sizeof(float);
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_bf16x6_kernel<1>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    (int)(FF_RING_BYTES + (3072 + 1536) * sizeof(float))) == hipSuccess;
    if (!attr_ok) return CTRLSIM_EINVAL;
    prof_before(PROF_GEMM, st);
    hipLaunchKernelGGL(ffn_fused_bf16x6_kernel<1>, dim3(grid), dim3(256), shm, st, O, ldo,
                       FfnPre{R, ldr, static_cast<const op_t*>(Wop), bo, g0, be0, nullptr, 0}, static_cast<const op_t*>(W1q), b1,
                       static_cast<const op_t*>(W2p), b2, gamma, beta, Y, ldy, M, F / 32, ctrlsim_nonfinite_ptr());
    // attention output and residual rows in, y out; Wo + W1 + W2 as NPL planes
    prof_after(PROF_GEMM, 4.0 * (double)M * DM * (double)F + 2.0 * (double)M * DM * DM, st,
               12.0 * (double)M * DM + 2.0 * NPL * (double)DM * DM + 4.0 * NPL * (double)DM * F, PKIND_FFN);
    return ctrlsim_launch_status();
  }
}

// x1 = LayerNorm0(R + Wo O + bo) and q = Wq x1 + bq as ONE kernel (QP above): the self-attention out-projection + residual + LayerNorm of a
// decoder layer and the cross-attention query projection behind it (nn.TransformerDecoderLayer: self_attn.out_proj + norm1, then the first
// third of multihead_attn.in_proj) — x1 is written once and not re-read.  Wop = pack.py:row_blocks(Wo), Wqp = pack.py:row_blocks of Wq with
// the k order of ffn_planes_pre.  X1 may alias R (or O).  Two-plane scheme only.
int launch_outproj_ln_q(const float* O, int ldo, const float* R, int ldr, const void* Wop, const float* bo, const float* g0, const float* be0,
                        const void* Wqp, const float* bq, float* X1, int ldx1, float* Q, int ldq, int M, hipStream_t st) {
  if (M <= 0) return CTRLSIM_OK;
  if (!O || !R || !Wop || !bo || !g0 || !be0 || !Wqp || !bq || !X1 || !Q || (ldo & 3) || (ldr & 3) || (ldx1 & 3) || (ldq & 3)) return CTRLSIM_EINVAL;
  if constexpr (!FF_PAIR) {
    return CTRLSIM_EINVAL;
  } else {
    const int n_rb = (M + 127) / 128;
    const int grid = n_rb < 256 ? n_rb : 256;
    const size_t shm = FF_RING_BYTES + (size_t)1024 * sizeof(float);
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_bf16x6_kernel<2>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FF_RING_BYTES + 1024 * sizeof(float))) == hipSuccess;
    if (!attr_ok) return CTRLSIM_EINVAL;
    prof_before(PROF_GEMM, st);
    hipLaunchKernelGGL(ffn_fused_bf16x6_kernel<2>, dim3(grid), dim3(256), shm, st, O, ldo,
                       FfnPre{R, ldr, static_cast<const op_t*>(Wop), bo, g0, be0, X1, ldx1}, static_cast<const op_t*>(Wqp), bq,
                       static_cast<const op_t*>(Wqp), bq, g0, be0, Q, ldq, M, 0, ctrlsim_nonfinite_ptr());
    // attention output and residual rows in, x1 and q out; Wo + Wq as NPL planes
    prof_after(PROF_GEMM, 4.0 * (double)M * DM * DM, st, 16.0 * (double)M * DM + 4.0 * NPL * (double)DM * DM, PKIND_GEMM_LN);
    return ctrlsim_launch_status();
  }
}

}  // namespace SPLIT_NS
