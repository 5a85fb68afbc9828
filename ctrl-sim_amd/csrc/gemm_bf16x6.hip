// fp32-accurate GEMM on the 16-bit MFMA: every fp32 operand is split into NPL 16-bit planes whose sum is the value and the
// product is evaluated as its leading partial products with fp32 accumulation inside the MFMA (csrc/split.h: two fp16 planes
// and three products by default — roof 833 TFLOP/s fp32-equivalent; three bf16 planes and six products — 417 — as the
// range-safe alternative; file and kernel names keep the historical "bf16x6").  The result carries fp32-class error (measured
// against an fp64 reference in tests/test_gpu_ops.py: same order as the f32-input MFMA kernel); token parity with the reference
// is unaffected (fixtures: bit-exact tokens, logits within 1e-4).
//
// Weights are split ONCE at pack time (ctrlsim_amd/pack.py, pre-scaled by WSCALE) into slab-major planes
// W3[K/16][NPL][2][N][8]  (the two 8-element halves of a 16-wide k-step are separate sub-planes) so a workgroup's K-slab of a
// sub-plane is one contiguous 16-byte-per-row stream and the LDS image [plane][half][row][8] makes every fragment read of a wave
// two contiguous 512-byte spans (no bank conflicts).  Activations stay fp32 in HBM and are split in registers while being
// staged into LDS (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32, round-to-nearest-even); the epilogue multiplies by 1 / WSCALE.
//
// Tiling: 256 threads = 4 waves, wave tile 64x64 (2x2 MFMA tiles, 64 accumulator registers), arranged
//   2x2 -> workgroup tile 128 x 128, 48 KB of LDS (three 16 KB stages; two 24 KB stages with three planes), THREE workgroups per CU
//   1x4 -> workgroup tile  64 x 256, 40 KB of LDS (60), three workgroups per CU                     (Linear + LayerNorm: whole rows)
// One LDS stage = one 16-wide k-step, one barrier per k-step; the 2x2 tiles keep THREE stages (W planes arrive by LDS-DMA two
// k-steps ahead), the 1x4 tiles two; the fp32 A rows are prefetched two k-steps ahead into alternating register sets by
// inline-asm loads with counted vmcnt waits (hipcc's own wait insertion drains everything: see gload_a / wait_a;
// tools/check_prefetch_regs.py checks the generated ISA).  Several independent workgroups per CU
// are what overlaps the phases: measured on the previous one-workgroup-per-CU version (8 waves in barrier lock-step)
// the k-loop, its operand staging and the epilogue simply added up (0.19 + 0.18 + 0.17 ms on the FFN-1 shape).
// Persistent workgroups with the XCD-aware tile order of gemm.hip; the next tile's first A slab is prefetched before
// the epilogue; the epilogue stages half the tile's rows at a time through LDS for 16-byte bias / residual / store
// traffic and, for LN, normalises whole rows there (one wave per row).
#define GEMM_NT_STORE   // fp32 output rows leave with nontemporal stores: they are far larger than L2 and only evict the operands (-2.5 %)
#define GEMM_RPRE_N 8      // residual rows requested ahead per chunk (all eight: four VGPRs spill, still 9 % faster than four ahead)
#define GEMM_RPRE    // LayerNorm epilogue: residual rows requested before the accumulators go through LDS (0.58 -> 0.49 ms, out-proj + LN shape)
#include "split.h"
#include <type_traits>

namespace SPLIT_NS {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define XK 16
// LDS stages of the k-loop: three for the 2x2 tiles of the two-plane scheme (16 KB per stage, 3 workgroups per CU = 144 KB);
// two where a third would cost a resident workgroup (1x4 tiles: 20 KB per stage; three planes: 24 KB)
#define GEMM_RING(WR_, WC_) ((NPL == 2 && (WR_) == 2 && (WC_) == 2) ? 3 : 2)

template <int PA, int PB, int MR>
__device__ __forceinline__ void term(f32x16 (&acc)[MR][2], const opx8 (&fa)[MR][NPL], const opx8 (&fb)[2][NPL]) {
#pragma unroll
  for (int a = 0; a < MR; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = MFMA_OP(fa[a][PA], fb[b][PB], acc[a][b]);
}

// KVIMG = true (2x2 tiles, plain Linear): output columns >= kv.k_col0 are attention keys (256 columns) and values (the
// next 256) and are written NOT as fp32 rows but directly as the split-bf16 K / V^T tile images the attention kernel
// stages by DMA (attention_bf16x6.hip: layout at kv_split_kernel) — the K/V split costs no extra pass over HBM.
// Up to MAXC classes of contexts share one launch (rows [row0, next row0) belong to class c): L rows per context, of which the rows
// >= Lreg (the representative tokens of a compact context, attention_bf16x6.hip) are keys rep_k0 + (row - Lreg), the others
// key = row; nkt tiles per (context, head), the class's images start at tile tile0.
struct KvClass { int row0, L, Lreg, rep_k0, nkt; long tile0; };
struct KvImg { op_t* img; int k_col0; int n; KvClass c[MAXC]; };
// (class tables are indexed with compile-time indices only: a run-time index into the by-value kernel argument makes the
// compiler copy the table to scratch memory)
__device__ __forceinline__ void kv_locate(const KvImg& kv, int grow, int& b, int& pos, int& nkt, long& tile0) {
  KvClass c = kv.c[0];
#pragma unroll
  for (int k = 1; k < MAXC; ++k)
    if (k < kv.n && grow >= kv.c[k].row0) c = kv.c[k];
  const int r = grow - c.row0;
  b = r / c.L;
  const int row = r - b * c.L;
  pos = row < c.Lreg ? row : c.rep_k0 + (row - c.Lreg);
  nkt = c.nkt;
  tile0 = c.tile0;
}

// The rows of an output tile are consecutive and almost always lie in ONE class: kv_tile() resolves the class, the context of
// the tile's first row and its row within that context once per tile with wave-uniform (scalar) arithmetic; a lane then places
// its row with an add and a compare instead of a class search and an integer division per row (the V branch of the epilogue
// calls this eight times per thread and chunk).  Tiles that straddle a class boundary, and classes whose contexts are shorter
// than a tile (the first steps of the K/V-cached phase), take kv_locate.
struct KvTile { bool fast; int b0, row_in0, L, Lreg, rep_k0, nkt; long tile0; };
__device__ __forceinline__ KvTile kv_tile(const KvImg& kv, int cbm, int rows) {
  KvClass c = kv.c[0];
  int next_row0 = kv.n > 1 ? kv.c[1].row0 : 0x7fffffff;
#pragma unroll
  for (int k = 1; k < MAXC; ++k)
    if (k < kv.n && cbm >= kv.c[k].row0) {
      c = kv.c[k];
      next_row0 = (k + 1 < MAXC && k + 1 < kv.n) ? kv.c[k + 1 < MAXC ? k + 1 : MAXC - 1].row0 : 0x7fffffff;
    }
  KvTile t;
  t.fast = cbm + rows - 1 < next_row0 && c.L >= rows;
  const int r0 = cbm - c.row0;
  const int b0 = r0 / c.L;
  // everything here is wave-uniform: pin it to scalar registers (the vector register file of this kernel is full)
  auto sc = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  t.fast = sc(t.fast ? 1 : 0) != 0;
  t.b0 = sc(b0);
  t.row_in0 = sc(r0 - b0 * c.L);
  t.L = sc(c.L); t.Lreg = sc(c.Lreg); t.rep_k0 = sc(c.rep_k0); t.nkt = sc(c.nkt);
  t.tile0 = ((long)sc((int)(c.tile0 >> 32)) << 32) | (unsigned)sc((int)(c.tile0 & 0xffffffff));
  return t;
}
__device__ __forceinline__ void kv_place(const KvImg& kv, const KvTile& t, int cbm, int grow, int& b, int& pos, int& nkt, long& tile0) {
  if (t.fast) {
    int row = t.row_in0 + (grow - cbm);
    const bool wrap = row >= t.L;
    b = t.b0 + (wrap ? 1 : 0);
    row -= wrap ? t.L : 0;
    pos = row < t.Lreg ? row : t.rep_k0 + (row - t.Lreg);
    nkt = t.nkt;
    tile0 = t.tile0;
  } else {
    kv_locate(kv, grow, b, pos, nkt, tile0);
  }
}

template <int WR, int WC, int MR, bool RELU, bool RESID, bool LN, bool KVIMG = false>
#define GEMM_OCC_22 3
#define GEMM_OCC_14 3        // 1x4 (LayerNorm) tiles: 40 KB of LDS with two planes -> three workgroups per CU (+24 % on out_proj+LN)
__global__ __launch_bounds__(256, MR == 1 ? 4 : ((WR == 2 && WC == 2) ? GEMM_OCC_22 : GEMM_OCC_14)) void gemm_nt_bf16x6_kernel(
    const float* __restrict__ A, int lda, const op_t* __restrict__ W3,   // [K/16][NPL][2][n_total][8]
    const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* R, int ldr, float* C, int ldc, int M, int N, int K, int m_tiles, int n_tiles,   // C may alias R: no restrict
    int n_total, int n0, KvImg kv, int* __restrict__ nonfinite) {
  // W3 holds all n_total rows of the packed matrix; this GEMM uses rows [n0, n0 + N) (e.g. the q / kv halves of an
  // in_proj_weight)
  constexpr int WMR = 32 * MR;                     // rows of a wave tile (MR x 2 MFMA tiles of 32 x 32)
  constexpr int XM = WMR * WR, XN = 64 * WC;
  constexpr int A_PLANE = XM * XK;                 // 16-bit elements of one plane of one k-step
  constexpr int W_PLANE = XN * XK;
  constexpr int STAGE = NPL * (A_PLANE + W_PLANE); // one k-step: 8 / 10 KB per plane (2x2 / 1x4)
  constexpr int CP = XN + 4, CR = 32 * WR;         // epilogue chunk: CR rows x XN columns of fp32
  constexpr int NA = (XM + 63) / 64;               // f32x4 loads of A per thread and stage
  constexpr int NW = 2 * NPL * XN / 256;           // 16-byte DMA chunks of W per thread and stage
  constexpr int RING = GEMM_RING(WR, WC);          // LDS stages: the W DMA runs RING - 1 k-steps ahead
  static_assert(!LN || (WR == 1 && WC == 4), "LayerNorm epilogue needs whole 256-wide rows");
  static_assert(XM % 64 == 0, "A staging assumes whole 64-row groups");
  extern __shared__ __attribute__((aligned(16))) op_t lds[];   // max(RING * STAGE elements, the epilogue chunk): see the launcher

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;

  const int total_ids = ((m_tiles + 7) / 8) * 8 * n_tiles;
  int bm = 0, bn = 0;
  auto tile_of = [&](int id, int& tbm, int& tbn) -> bool {
    const int xcd = id & 7, j = id >> 3;
    const int mt = (j / n_tiles) * 8 + xcd, nt = j % n_tiles;
    tbm = mt * XM;
    tbn = nt * XN;
    return mt < m_tiles;
  };

  // ---- staging: A through registers (fp32 -> 3 bf16 planes), W planes by LDS-DMA (global_load_lds, 16 B per lane:
  // the W part of a stage is lane-linear in exactly the order idx = tid + 256*i, so the DMA needs no VGPRs at all)
  // A is prefetched TWO k-steps ahead into alternating register sets (the fp32 rows come from HBM: one k-step of MFMAs does
  // not cover that latency), W RING - 1 k-steps ahead.  All loads are unconditional (rows / columns past the edge are clamped:
  // they are computed and never stored) so that the in-order vmcnt arithmetic of the k-loop is exact.
  f32x4 ra[2][NA];
  const size_t w_slab = (size_t)NPL * n_total * XK;  // elements per 16-wide K-slab of W3
  auto gload_a = [&](int kt, auto SET) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + 256 * i, r = idx >> 2, c = (idx & 3) * 4;
      int ga = bm + r;
      ga = ga < M ? ga : M - 1;
      // inline asm: hipcc's own s_waitcnt insertion answers a register load that is consumed two k-steps later with vmcnt(0)
      // (draining the prefetches behind it); loads it does not see are waited for by the counted wait_a() below instead
      const float* src = A + (size_t)ga * lda + kt * XK + c;
      f32x4 t;                                     // (a local: clang rejects captured arrays as asm operands in a generic lambda)
      asm volatile("global_load_dwordx4 %0, %1, off ; A-PREFETCH" : "=v"(t) : "v"(src) : "memory");
      ra[set][i] = t;
    }
  };
  auto dma_w = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = tid + 256 * i, pc = idx / XN, r = idx % XN;
      int gw = bn + r;
      gw = gw < N ? gw : N - 1;                    // columns >= N are computed on clamped rows and never stored
      const op_t* src = W3 + (size_t)kt * w_slab + ((size_t)pc * n_total + n0 + gw) * 8;
      op_t* dst = lds + buf * STAGE + NPL * A_PLANE + (wave * 64 + 256 * i) * 8;   // wave-uniform LDS base
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  // the register set becomes valid once at most NEWER younger VMEM operations are in flight (in-order return); the "+v" ties keep
  // every use of the set behind the wait
  auto wait_a = [&](auto SET, auto NEWER) {
    constexpr int set = decltype(SET)::value, newer = decltype(NEWER)::value;
    static_assert(newer < 64, "vmcnt is a 6-bit counter");
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      f32x4 t = ra[set][i];
      if (i == 0) asm volatile("s_waitcnt vmcnt(%1) ; A-WAIT" : "+v"(t) : "n"(newer) : "memory");
      else asm volatile("" : "+v"(t));
      ra[set][i] = t;
    }
  };
  auto sstore_a = [&](auto SET, int buf) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + 256 * i, r = idx >> 2, c = (idx & 3) * 4;
      u32x2 pl[NPL];
      split_quad(ra[set][i], pl);
      op_t* Ab = lds + buf * STAGE + (c >> 3) * (A_PLANE / 2) + r * 8 + (c & 7);
#pragma unroll
      for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(Ab + q * A_PLANE) = pl[q];
    }
  };
  // end of a k-step: this wave's LDS writes are complete (lgkmcnt(0)), every VMEM operation except the newest `keep` ones has
  // returned (vmcnt counts in order; gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt(7) = no wait), then the barrier
  auto step_barrier = [&](bool newer_in_flight) {
    constexpr int keep = (RING == 3 ? NW : 0) + NA;
    asm volatile("" ::: "memory");
    if (newer_in_flight) __builtin_amdgcn_s_waitcnt(0x0070 | (keep & 15) | ((keep >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  using set0 = std::integral_constant<int, 0>;
  using set1 = std::integral_constant<int, 1>;

  const int nk = K / XK;
  int id = blockIdx.x;
  while (id < total_ids && !tile_of(id, bm, bn)) id += gridDim.x;
  if (id >= total_ids) return;
  gload_a(0, set0{});
  for (;;) {
    f32x16 acc[MR][2];
#pragma unroll
    for (int a = 0; a < MR; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // one k-step on stage `cur`; SET = register set that receives A of step kt + 2 (it held step kt, staged one step ago)
    auto kstep = [&](int kt, int cur, auto SET) {
      constexpr int set = decltype(SET)::value;
      using other = std::integral_constant<int, set ^ 1>;
      const int nxt = cur + 1 == RING ? 0 : cur + 1;
      // the prefetches are issued UNCONDITIONALLY (past the end of K they re-fetch the last slab into a released stage / a dead
      // register set): a conditional issue makes the number of operations in flight path dependent, and both hipcc's own
      // vmcnt for the register set and the counted wait below would have to assume the worst (vmcnt(0): no prefetch at all)
      const int kpre = kt + 2 < nk ? kt + 2 : nk - 1;
      if (RING == 3) dma_w(kpre, nxt + 1 == RING ? 0 : nxt + 1);     // stage of step kt - 1: released by its barrier
      else dma_w(kt + 1 < nk ? kt + 1 : nk - 1, nxt);
      gload_a(kpre, SET);
      {
        const op_t* Ab = lds + cur * STAGE + half * (A_PLANE / 2) + (wr * WMR + l31) * 8;
        const op_t* Wb = lds + cur * STAGE + NPL * A_PLANE + half * (W_PLANE / 2) + (wc * 64 + l31) * 8;
        opx8 fa[MR][NPL], fb[2][NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
          for (int a = 0; a < MR; ++a) fa[a][p] = *reinterpret_cast<const opx8*>(Ab + p * A_PLANE + a * 32 * 8);
#pragma unroll
          for (int b = 0; b < 2; ++b) fb[b][p] = *reinterpret_cast<const opx8*>(Wb + p * W_PLANE + b * 32 * 8);
        }
        // the partial products, smallest first; term-major order keeps 4 independent accumulators between reuses
#if CTRLSIM_F16X3
        term<1, 0, MR>(acc, fa, fb);
        term<0, 1, MR>(acc, fa, fb);
        term<0, 0, MR>(acc, fa, fb);
#else
        term<2, 0, MR>(acc, fa, fb);
        term<0, 2, MR>(acc, fa, fb);
        term<1, 1, MR>(acc, fa, fb);
        term<1, 0, MR>(acc, fa, fb);
        term<0, 1, MR>(acc, fa, fb);
        term<0, 0, MR>(acc, fa, fb);
#endif
      }
      wait_a(other{}, std::integral_constant<int, NW + NA>{});   // younger: this step's two prefetches
      sstore_a(other{}, nxt);                      // A of step kt + 1 (loaded during step kt - 1)
      // W of step kt + 1 must have landed: with three stages it was issued BEFORE the A rows just consumed (in-order return);
      // with two it is the oldest operation of this step.  Only this step's prefetches may stay in flight — none after the
      // last step: the epilogue reuses the stages.
      step_barrier(kt + 1 < nk);
    };

    dma_w(0, 0);
    if (RING == 3) dma_w(nk > 1 ? 1 : 0, 1);
    gload_a(nk > 1 ? 1 : 0, set1{});
    wait_a(set0{}, std::integral_constant<int, (RING - 1) * NW + NA>{});
    sstore_a(set0{}, 0);
    step_barrier(true);
    {
      int cur = 0;
      for (int kt = 0; kt < nk; kt += 2) {
        kstep(kt, cur, set0{});
        cur = cur + 1 == RING ? 0 : cur + 1;
        if (kt + 1 < nk) {
          kstep(kt + 1, cur, set1{});
          cur = cur + 1 == RING ? 0 : cur + 1;
        }
      }
    }

    // ---- epilogue: two chunks of CR rows (MFMA tile row a of every wave row) staged through LDS
    float* Cs = reinterpret_cast<float*>(lds);   // [CR][XN + 4]
    const bool vec_ok = !(ldc & 3) && (!RESID || !(ldr & 3));
    const int cbm = bm, cbn = bn;
    int nid = id + gridDim.x;
    while (nid < total_ids && !tile_of(nid, bm, bn)) nid += gridDim.x;
    const bool have_next = nid < total_ids;
    if (have_next) gload_a(0, set0{});
#pragma unroll
    for (int a = 0; a < MR; ++a) {
      // LayerNorm epilogue: the residual rows of this chunk are requested BEFORE the accumulators go through LDS, so their
      // HBM latency runs under the ds_write / barrier instead of in front of the row reductions
      constexpr int NRP_ALL = CR * (XN / 4) / 256;
      constexpr int NRP = (LN && RESID) ? (NRP_ALL < GEMM_RPRE_N ? NRP_ALL : GEMM_RPRE_N) : 1;   // rows requested ahead (the rest in the loop: register budget)
      f32x4 rpre[NRP];
      if (LN && RESID) {
#pragma unroll
        for (int i = 0; i < NRP; ++i) {
          const int idx = tid + 256 * i, lr = idx / (XN / 4), col = (idx % (XN / 4)) * 4;
          const int grow = cbm + (lr >> 5) * WMR + a * 32 + (lr & 31);
          rpre[i] = grow < M ? *reinterpret_cast<const f32x4*>(R + (size_t)grow * ldr + cbn + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[(wr * 32 + mfma_row(r, half)) * CP + wc * 64 + b * 32 + l31] = acc[a][b][r] * WSCALE_INV;
      __syncthreads();
      if (KVIMG && cbn >= kv.k_col0) {
        // ---- this tile is 4 heads of K or of V: emit the images.  Chunk rows lr = s*32 + j <-> global row
        // cbm + s*64 + a*32 + j (two 32-row segments); key position = row % L, context = row / L.
        constexpr int KIMG = 2 * NPL * 64 * HD, KPL = 64 * HD;    // image / plane sizes in 16-bit elements
        const int rel = cbn - kv.k_col0;
        const bool isV = rel >= DM;
        const KvTile kt_ = kv_tile(kv, __builtin_amdgcn_readfirstlane(cbm), XM);
        const int head0 = (rel & (DM - 1)) >> 5;
        if (!isV) {
          const int lr = tid & 63;
          const int grow = cbm + (lr >> 5) * WMR + a * 32 + (lr & 31);
          if (grow < M) {
            int b, pos, nkt;
            long tile0;
            kv_place(kv, kt_, cbm, grow, b, pos, nkt, tile0);
            const int kt = pos >> 6, key = pos & 63;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int c = (tid >> 6) + 4 * i, hh = c >> 2, dg = c & 3;     // 8 dims [8c, 8c+8) of the tile's 128 columns
              f32x4 x0 = *reinterpret_cast<const f32x4*>(Cs + lr * CP + c * 8);
              f32x4 x1 = *reinterpret_cast<const f32x4*>(Cs + lr * CP + c * 8 + 4);
              if (bias) {
                x0 += *reinterpret_cast<const f32x4*>(bias + cbn + c * 8);
                x1 += *reinterpret_cast<const f32x4*>(bias + cbn + c * 8 + 4);
              }
              u32x2 pa[NPL], pb[NPL];
              split_quad(x0, pa);
              split_quad(x1, pb);
              op_t* dst = kv.img + (tile0 + ((size_t)b * NHEAD + head0 + hh) * nkt + kt) * KIMG + (dg * 64 + key) * 8;
#pragma unroll
              for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x4*>(dst + q * KPL) = u32x4{pa[q][0], pa[q][1], pb[q][0], pb[q][1]};
            }
          }
        } else {
          const int col = tid & 127, hh = col >> 5, d = col & 31;
          const float bv = bias ? bias[cbn + col] : 0.f;
#pragma unroll
          for (int i = 0; i < CR / 8; ++i) {
            const int lr0 = ((tid >> 7) + 2 * i) * 4;                      // a key quad: 4 consecutive rows of one segment
            const int grow0 = cbm + (lr0 >> 5) * WMR + a * 32 + (lr0 & 31);
            if (grow0 < M) {                                               // L % 4 == Lreg % 4 == 0: quads never straddle contexts / regions
              int b, pos, nkt;
              long tile0;
              kv_place(kv, kt_, cbm, grow0, b, pos, nkt, tile0);
              const int kt = pos >> 6, q = (pos & 63) >> 2;
              const float x0 = Cs[(lr0 + 0) * CP + col] + bv, x1 = Cs[(lr0 + 1) * CP + col] + bv;
              const float x2 = Cs[(lr0 + 2) * CP + col] + bv, x3 = Cs[(lr0 + 3) * CP + col] + bv;
              u32x2 pv[NPL];
              split_quad(f32x4{x0, x1, x2, x3}, pv);
              op_t* dst = kv.img + (tile0 + ((size_t)b * NHEAD + head0 + hh) * nkt + kt) * KIMG + NPL * KPL + (q * HD + d) * 4;
#pragma unroll
              for (int qq = 0; qq < NPL; ++qq) *reinterpret_cast<u32x2*>(dst + qq * KPL) = pv[qq];
            }
          }
        }
        __syncthreads();
        continue;
      }
      constexpr int LPR = XN / 4;                  // lanes per row (f32x4 each): 32 (2x2) or 64 = one wave (1x4)
#pragma unroll
      for (int i = 0; i < CR * LPR / 256; ++i) {
        const int idx = tid + 256 * i, lr = idx / LPR, col = (idx % LPR) * 4;
        const int grow = cbm + (lr >> 5) * WMR + a * 32 + (lr & 31), gcol = cbn + col;
        if (LN) {
          // one wave = one full 256-wide row (lane -> 4 consecutive columns); rows beyond M are skipped wave-uniformly
          if (grow >= M) continue;
          f32x4 v = *reinterpret_cast<const f32x4*>(Cs + lr * CP + col);
          if (bias) v += *reinterpret_cast<const f32x4*>(bias + gcol);
          if (RESID) v += i < NRP ? rpre[i < NRP ? i : 0] : *reinterpret_cast<const f32x4*>(R + (size_t)grow * ldr + gcol);
          const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
          const f32x4 dv = v - mean;
          const float var = wave_sum(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2] + dv[3] * dv[3]) * (1.f / 256.f);
          count_nonfinite_row(var, lane, nonfinite);
          f32x4 y = dv * (1.0f / sqrtf(var + 1e-5f)) * *reinterpret_cast<const f32x4*>(gamma + gcol) +
                    *reinterpret_cast<const f32x4*>(beta + gcol);
          if (RELU) {
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = fmaxf(y[c], 0.f);
          }
          __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(C + (size_t)grow * ldc + gcol));
          continue;
        }
        if (grow >= M || gcol >= N) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(Cs + lr * CP + col);
        if (vec_ok && gcol + 3 < N) {
          if (bias) v += *reinterpret_cast<const f32x4*>(bias + gcol);
          if (RESID) v += *reinterpret_cast<const f32x4*>(R + (size_t)grow * ldr + gcol);
          if (RELU) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
          }
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(C + (size_t)grow * ldc + gcol));
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (gcol + c < N) {
              float x = v[c] + (bias ? bias[gcol + c] : 0.f);
              if (RESID) x += R[(size_t)grow * ldr + gcol + c];
              if (RELU) x = fmaxf(x, 0.f);
              C[(size_t)grow * ldc + gcol + c] = x;
            }
          }
        }
      }
      __syncthreads();
    }
    if (!have_next) break;
    id = nid;
  }
}

int launch_gemm_nt_bf16x6_kvc(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias,
                              const float* R, int ldr, float* C, int ldc, int M, int N, int K, int relu,
                              const float* ln_gamma, const float* ln_beta, void* kv_img, int kv_col0, int kv_n,
                              const KvClassHost* kv_cls, hipStream_t st);
int launch_gemm_nt_bf16x6(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias,
                          const float* R, int ldr, float* C, int ldc, int M, int N, int K, int relu,
                          const float* ln_gamma, const float* ln_beta, hipStream_t st) {
  return launch_gemm_nt_bf16x6_kvc(A, lda, W3, n_total, n0, bias, R, ldr, C, ldc, M, N, K, relu, ln_gamma, ln_beta, nullptr, 0, 0,
                                   nullptr, st);
}
// kv_img != NULL: columns [kv_col0, kv_col0 + 512) are keys / values of 8 heads x 32 and go to the split images of
// kv_nkt 64-key tiles per context of kv_L rows (kv_L % 4 == 0, kv_L >= 32, kv_col0 % 128 == 0, M % kv_L == 0); rows >= kv_Lreg of a
// context (kv_Lreg % 4 == 0; kv_Lreg = kv_L: none) are written from key kv_rep_k0 (a multiple of 64) on
int launch_gemm_nt_bf16x6_kv(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias,
                             const float* R, int ldr, float* C, int ldc, int M, int N, int K, int relu,
                             const float* ln_gamma, const float* ln_beta, void* kv_img, int kv_L, int kv_nkt, int kv_col0,
                             int kv_Lreg, int kv_rep_k0, hipStream_t st) {
  if (!kv_img) return launch_gemm_nt_bf16x6(A, lda, W3, n_total, n0, bias, R, ldr, C, ldc, M, N, K, relu, ln_gamma, ln_beta, st);
  if (kv_L <= 0 || M % kv_L) return CTRLSIM_EINVAL;
  if (kv_Lreg <= 0) { kv_Lreg = kv_L; kv_rep_k0 = 0; }
  const KvClassHost c{M / kv_L, kv_L, kv_Lreg, kv_rep_k0, kv_nkt, 0};
  return launch_gemm_nt_bf16x6_kvc(A, lda, W3, n_total, n0, bias, R, ldr, C, ldc, M, N, K, relu, ln_gamma, ln_beta, kv_img, kv_col0,
                                   1, &c, st);
}

#if CTRLSIM_F16X3
// ---- Weight-stationary Linear(256 -> 256) [+ residual] [+ LayerNorm] [+ ReLU] for tall row matrices (OPT_GEMM_WS).
// The tiled kernel above re-streams the 256 KB of weight planes for every 64-row tile and keeps three phases per k-step in
// barrier lock-step; on this shape (3 KB of HBM traffic per row against 0.26 MFLOP) it reaches about 60 % of its HBM floor.
// Here the WEIGHTS stay in registers for the life of a persistent workgroup: 8 waves, wave w owns output columns
// [32 w, 32 w + 32) and holds their 16 k-steps x 2 planes as MFMA A-operand fragments (128 VGPRs).  Rows stream through in
// blocks of 32, and every byte of global traffic is a whole 1 KB row per instruction:
//   * activation rows AND residual rows arrive as raw fp32 by LDS-DMA (padded LDS rows: conflict-free 16-byte fragment reads),
//     requested two blocks ahead (two buffers each), so a block's loads have a whole iteration to land;
//   * every wave splits the activation rows it reads in registers (the same values in all 8 waves: 12 VALU instructions per
//     k-step beside the 3 MFMAs) and computes D^T = W_slab . A^T, so a lane ends up with 16 columns of ONE row: bias /
//     residual / LayerNorm / ReLU run in registers; the per-wave LayerNorm partials (mean and centred sum of squares of 32
//     columns) are merged with the parallel-variance formula after the block's first barrier;
//   * results are written over the residual rows in LDS (same lane, same addresses) and leave, after the second barrier, as
//     whole rows (16 bytes per lane, nontemporal).
// The first barrier is taken after `s_waitcnt vmcnt(0)`: it also tells every wave that the next block's rows have landed and
// that the activation buffer the k-loop just read may be refilled.  The DMA is issued as inline asm (see
// attention_bf16x6.hip: the compiler answers the builtin with a full drain before the next ds_read).
#define WS_ROWS 32
#define WS_AHEAD 2
__device__ __forceinline__ f32x16 ws_fake_mfma(opx8 a, opx8 b, f32x16 c) {      // ablation builds only
  c[0] += (float)a[0] * (float)b[0];
  return c;
}
// eight fp32 values -> hi / residual fp16 fragments; the residual straight from the mixed-precision FMA (lo = f16(a - (float)hi):
// the same bits as convert - subtract - convert, half the instructions)
__device__ __forceinline__ void ws_split8(const f32x4 a, const f32x4 b, opx8 (&f)[NPL]) {
  u32x4 hi, lo;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = i < 2 ? a[2 * i] : b[2 * i - 4], x1 = i < 2 ? a[2 * i + 1] : b[2 * i - 3];
    const unsigned h = op_cvt_pk(x0, x1);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x1));
    hi[i] = h; lo[i] = l;
  }
  f[0] = __builtin_bit_cast(opx8, hi);
  f[1] = __builtin_bit_cast(opx8, lo);
}
#define WS_LD 260          // floats per LDS row (1040 B: the 16 lanes of a ds_read_b128 group hit 16 different bank quads)
#define WS_LDS_BYTES ((4 * WS_ROWS * WS_LD + 3 * 256 + 8 * WS_ROWS * 2) * 4)
// KV = true (plain Linear, N = 256 G, one column group of 256 per blockIdx.y: the workgroups of group 0 are dispatched first,
// the others as compute units come free): groups at or beyond kv.k_col0 are the attention keys, then the values, and leave as
// the split K / V^T tile images (layout and class tables: see the tiled kernel above) straight from the staged result rows.
template <bool RELU, bool RESID, bool LN, bool KV = false>
__global__ __launch_bounds__(512, 2) void gemm_ws256_kernel(const float* A, int lda, const op_t* __restrict__ W3,
                                                            const float* __restrict__ bias, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* R, int ldr, float* C,
                                                            int ldc, int M, int n_total, int n0, int* __restrict__ nonfinite,
                                                            const KvImg kv, const int* __restrict__ c_rows = nullptr) {
  // c_rows (row-store epilogue only): result row i leaves as row c_rows[i] of C (< 0: not stored) — a Linear whose output rows are
  // scattered into a larger row space (the map encoder's last Linear writes the polyline rows of the scene-encoder source) needs no
  // copy kernel behind it.  The index is read with a SCALAR load (the row is wave-uniform): the counted vmcnt waits below are unchanged.
  static_assert(!KV || (!RELU && !RESID && !LN), "the K / V image epilogue belongs to the plain Linear");
  const int grp = KV ? (int)blockIdx.y : 0;
  n0 += 256 * grp;
  if (KV && bias) bias += 256 * grp;
  const int kv_kind = !KV ? 0 : (256 * grp < kv.k_col0 ? 0 : (256 * grp == kv.k_col0 ? 1 : 2));    // 0 = fp32 rows, 1 = keys, 2 = values
  if (KV) C += 256 * grp;
  extern __shared__ __attribute__((aligned(16))) float ws_lds[];
  float* const abuf = ws_lds;                               // [2][WS_ROWS * WS_LD]  activation rows
  float* const rbuf = ws_lds + 2 * WS_ROWS * WS_LD;         // [2][WS_ROWS * WS_LD]  residual rows in, result rows out
  float* const cvec = ws_lds + 4 * WS_ROWS * WS_LD;         // [3][256]              bias, gamma, beta
  float* const part = cvec + 3 * 256;                       // [8][WS_ROWS][2]       LayerNorm partials
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int nblk = (M + WS_ROWS - 1) / WS_ROWS;
  if ((int)blockIdx.x >= nblk) return;

  // rows 4 wave .. 4 wave + 3 of block (blockIdx.x + j * gridDim.x) into buffer j & 1
  auto dma = [&](const float* src0, int ld, float* dst0, int j) {
    const int blk = blockIdx.x + j * gridDim.x;
    if (blk >= nblk) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * wave + q;
      int grow = blk * WS_ROWS + r;
      grow = grow < M ? grow : M - 1;                                          // rows past the end: any valid row (never stored)
      const float* src = src0 + (size_t)grow * ld + lane * 4;
      const unsigned lds_addr = (unsigned)(size_t)((__attribute__((address_space(3))) float*)(dst0 + (j & 1) * WS_ROWS * WS_LD + r * WS_LD));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off ; WS-DMA"
                   :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(src) : "memory");
    }
  };
  dma(A, lda, abuf, 0);
  if (RESID) dma(R, ldr, rbuf, 0);

  opx8 wf[16][NPL];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int p = 0; p < NPL; ++p)
      wf[ks][p] = *reinterpret_cast<const opx8*>(W3 + ((((size_t)ks * NPL + p) * 2 + half) * n_total + n0 + 32 * wave + l31) * 8);
  if (tid < 256) {
    cvec[tid] = bias ? bias[tid] : 0.f;
    cvec[256 + tid] = LN ? gamma[tid] : 0.f;
    cvec[512 + tid] = LN ? beta[tid] : 0.f;
  }
  dma(A, lda, abuf, 1);
  if (RESID) dma(R, ldr, rbuf, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int c0 = 32 * wave + 4 * half;            // this lane's columns: c0 + 8 q + j
  const int njobs = (nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const float* const ap0 = abuf + l31 * WS_LD + 8 * half;
  const float* const cv = cvec + c0;

  // LayerNorm / ReLU of block `it` (values v, a lane's 16 columns of one row), cut into 16 slots that ride in the VALU gaps of
  // the NEXT block's k-loop (slot s runs between the MFMAs of k-step s).  Slots 0-3: merge the eight per-wave partials
  // (parallel-variance formula); 4-11: two columns each; 12-15: one result quad each into the row's place in rb.
  float v[16];
  f32x2 pw[8];
  f32x2 gq, bq;
  float mean = 0.f, rstd = 1.f, nmr = 0.f;
  auto epi_slot = [&](int s, float* yrow, bool live_row) {
    if (LN) {
      if (s == 0) {
#pragma unroll
        for (int w = 0; w < 8; ++w) pw[w] = *reinterpret_cast<const f32x2*>(part + (w * WS_ROWS + l31) * 2);
      } else if (s == 1) {
        mean = ((pw[0][0] + pw[1][0]) + (pw[2][0] + pw[3][0])) + ((pw[4][0] + pw[5][0]) + (pw[6][0] + pw[7][0]));
        mean *= (1.f / 8.f);
      } else if (s == 2) {
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) m2 += pw[w][1] + 32.f * (pw[w][0] - mean) * (pw[w][0] - mean);
        rstd = m2 * (1.f / 256.f);                       // the variance, until slot 3
      } else if (s == 3) {
        if (wave == 0 && half == 0 && live_row && !(rstd <= 3.0e38f)) atomicAdd(nonfinite, 1);
        rstd = 1.0f / sqrtf(rstd + 1e-5f);
        nmr = -mean * rstd;
        gq = *reinterpret_cast<const f32x2*>(cv + 256);
        bq = *reinterpret_cast<const f32x2*>(cv + 512);
      } else if (s < 12) {
        const int e = 2 * (s - 4);                       // columns e, e + 1 of the lane's 16: c0 + 8 (e >> 2) + (e & 3)
        const f32x2 g2 = gq, b2 = bq;
        if (s < 11) {
          const int en = e + 2, off = 8 * (en >> 2) + (en & 3);
          gq = *reinterpret_cast<const f32x2*>(cv + 256 + off);
          bq = *reinterpret_cast<const f32x2*>(cv + 512 + off);
        }
        v[e] = (v[e] * rstd + nmr) * g2[0] + b2[0];
        v[e + 1] = (v[e + 1] * rstd + nmr) * g2[1] + b2[1];
      }
    }
    if (s >= 12) {
      const int q = s - 12;
      f32x4 y = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      if (RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = fmaxf(y[j], 0.f);
      }
      *reinterpret_cast<f32x4*>(yrow + 8 * q) = y;
    }
  };

  // one block's 48 MFMAs; fragment reads run WS_AHEAD k-steps ahead (two waves per SIMD do not hide an LDS round trip per
  // k-step on their own; the scheduling barriers keep the compiler from sinking the reads back to their use).  EPI: the
  // previous block's epilogue slots ride along.
  f32x16 acc, acc1;
  auto kloop = [&](const float* ap, auto epi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
    f32x4 xq[WS_AHEAD + 1][2];
#pragma unroll
    for (int ks = 0; ks < WS_AHEAD; ++ks) {
      xq[ks][0] = *reinterpret_cast<const f32x4*>(ap + ks * 16);
      xq[ks][1] = *reinterpret_cast<const f32x4*>(ap + ks * 16 + 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const f32x4 lo = xq[ks % (WS_AHEAD + 1)][0], hi = xq[ks % (WS_AHEAD + 1)][1];
      opx8 fb[NPL];
      ws_split8(lo, hi, fb);
#define WS_MFMA(A_, B_, C_) MFMA_OP(A_, B_, C_)
      acc1 = WS_MFMA(wf[ks][1], fb[0], acc1);            // the two small products share an accumulator, W_hi x_hi has its own:
      acc = WS_MFMA(wf[ks][0], fb[0], acc);              // no MFMA waits for the one issued just before it
      if (ks + WS_AHEAD < 16) {
        xq[(ks + WS_AHEAD) % (WS_AHEAD + 1)][0] = *reinterpret_cast<const f32x4*>(ap + (ks + WS_AHEAD) * 16);
        xq[(ks + WS_AHEAD) % (WS_AHEAD + 1)][1] = *reinterpret_cast<const f32x4*>(ap + (ks + WS_AHEAD) * 16 + 4);
      }
      epi(ks);
      acc1 = WS_MFMA(wf[ks][0], fb[1], acc1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  kloop(ap0, [](int) {});

  for (int it = 0; it < njobs; ++it) {
    const int blk = blockIdx.x + it * gridDim.x;
    float* rb = rbuf + (it & 1) * WS_ROWS * WS_LD;
    float* yrow = rb + l31 * WS_LD + c0;
    const bool live_row = blk * WS_ROWS + l31 < M;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 t = *reinterpret_cast<const f32x4*>(cv + 8 * q);
      if (RESID) t += *reinterpret_cast<const f32x4*>(yrow + 8 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * q + j] = (acc1[4 * q + j] + acc[4 * q + j]) * WSCALE_INV + t[j];
    }
    if (LN) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += v[r];
      s += __shfl_xor(s, 32);
      const float mw = s * (1.f / 32.f);
      float m2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) m2 += (v[r] - mw) * (v[r] - mw);
      m2 += __shfl_xor(m2, 32);
      if (half == 0) *reinterpret_cast<f32x2*>(part + (wave * WS_ROWS + l31) * 2) = f32x2{mw, m2};
    }
    // Requests of this wave still in flight, oldest first: activation rows of block it + 1 (4), result rows of block it - 1 (4),
    // residual rows of block it + 1 (4).  Only the first four are needed now: the others stay in flight across the barrier
    // (vmcnt counts in issue order; raw s_barrier: __syncthreads would drain the counter).
    // The constants follow from the issue counts per wave and block, all UNCONDITIONAL so that the count is exact: dma() issues 4 row
    // requests (one per q), the row-store epilogue 4 stores (one per q), the key-image epilogue 2 x NPL = 4 stores, the value-image
    // epilogue 4 x NPL = 8 stores.  A lane whose row lies beyond M still issues its dma (clamped row) but skips its stores — that happens
    // only in the matrix's last, partial block, which is the LAST job of its workgroup (blocks are dealt round-robin), so no later wait
    // counts on those stores.
    static_assert(NPL == 2 && WS_ROWS == 32, "the counted vmcnt waits below assume 4 requests per dma(), 4 / 4 / 8 stores per epilogue");
    if (RESID) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (KV && kv_kind == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");     // a value block leaves as 8 stores per lane
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // block it + 1 has landed for every wave; partials visible
    dma(A, lda, abuf, it + 2);                            // into the buffer block it was computed from (before the last barrier)
    if (it + 1 < njobs) kloop(ap0 + ((it + 1) & 1) * WS_ROWS * WS_LD, [&](int s) { epi_slot(s, yrow, live_row); });
    else {
#pragma unroll
      for (int s = 0; s < 16; ++s) epi_slot(s, yrow, live_row);
    }
    if (RESID) {                                          // residual rows of block it + 1 (younger: the 4 activation rows just requested, if any)
      if (it + 2 < njobs) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                         // result rows complete in rb
    if (KV && kv_kind != 0) {
      constexpr int KIMG = 2 * NPL * 64 * HD, KPL = 64 * HD;    // image / plane sizes in 16-bit elements
      const int cbm = blk * WS_ROWS;
      const KvTile kt_ = kv_tile(kv, __builtin_amdgcn_readfirstlane(cbm), WS_ROWS);
      if (kv_kind == 1) {
        // keys: lane -> row tid & 31 and two of the tile's 32 groups of 8 dims (head c >> 2, dim group c & 3); 16 bytes per plane
        const int r = tid & 31, grow = cbm + r;
        if (grow < M) {
          int b, pos, nkt;
          long tile0;
          kv_place(kv, kt_, cbm, grow, b, pos, nkt, tile0);
          const int kt = pos >> 6, key = pos & 63;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int c = (tid >> 5) + 16 * i, hh = c >> 2, dg = c & 3;
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(rb + r * WS_LD + c * 8);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(rb + r * WS_LD + c * 8 + 4);
            u32x2 pa[NPL], pb[NPL];
            split_quad(x0, pa);
            split_quad(x1, pb);
            op_t* dst = kv.img + (tile0 + ((size_t)b * NHEAD + hh) * nkt + kt) * KIMG + (dg * 64 + key) * 8;
#pragma unroll
            for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x4*>(dst + q * KPL) = u32x4{pa[q][0], pa[q][1], pb[q][0], pb[q][1]};
          }
        }
      } else {
        // values: lane -> column tid & 255 (head col >> 5, dim col & 31) and four of the block's 8 key quads; 8 bytes per plane
        const int col = tid & 255, hh = col >> 5, d = col & 31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int lr0 = ((tid >> 8) + 2 * i) * 4, grow0 = cbm + lr0;
          if (grow0 < M) {                                               // L % 4 == Lreg % 4 == 0: quads never straddle contexts / regions
            int b, pos, nkt;
            long tile0;
            kv_place(kv, kt_, cbm, grow0, b, pos, nkt, tile0);
            const int kt = pos >> 6, q = (pos & 63) >> 2;
            const f32x4 x = {rb[(lr0 + 0) * WS_LD + col], rb[(lr0 + 1) * WS_LD + col], rb[(lr0 + 2) * WS_LD + col], rb[(lr0 + 3) * WS_LD + col]};
            u32x2 pv[NPL];
            split_quad(x, pv);
            op_t* dst = kv.img + (tile0 + ((size_t)b * NHEAD + hh) * nkt + kt) * KIMG + NPL * KPL + (q * HD + d) * 4;
#pragma unroll
            for (int qq = 0; qq < NPL; ++qq) *reinterpret_cast<u32x2*>(dst + qq * KPL) = pv[qq];
          }
        }
      }
      continue;
    }
    f32x4 yo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) yo[q] = *reinterpret_cast<const f32x4*>(rb + (4 * wave + q) * WS_LD + lane * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int grow = __builtin_amdgcn_readfirstlane(blk * WS_ROWS + 4 * wave + q);
      if (grow < M) {
        int orow = grow;
        if (c_rows) orow = ((const __attribute__((address_space(4))) int*)c_rows)[grow];
        if (orow >= 0) __builtin_nontemporal_store(yo[q], reinterpret_cast<f32x4*>(C + (size_t)orow * ldc + lane * 4));
      }
    }
    if (RESID) dma(R, ldr, rbuf, it + 2);                 // this wave's four rows of rb: read by its own stores above only
  }
}
#endif

#if CTRLSIM_F16X3
// ---- Row-stationary Linear(256 -> 32 nb) whose key / value columns leave as K / V^T tile images (OPT_GEMM_WS bit 3; round 4).
// The weight-stationary kernel above reads every activation row once per 256-column group and keeps the one resident workgroup of a
// CU in matrix / vector / memory lock-step (0.26 of the split roof on the in_proj).  Here the roles are those of the fused feed-forward
// kernel's first product: a wave keeps 32 ROWS as split operand fragments in registers (128 VGPRs) for the life of a 256-row job, the
// weights stream through a four-slot LDS ring by LDS-DMA as 32 KB blocks of 32 output columns (pack.py:row_blocks, the layout of the
// FFN's W1 blocks; 3 KB of L2 reads per row, shared by the 8 waves), and every block is 48 MFMAs per wave into a fresh accumulator:
//   * activation rows are read from HBM ONCE (1 KB per row instead of 1 KB per column group);
//   * 8 waves per CU = two per SIMD with independent accumulator chains: one wave's epilogue (split + stores) runs beside the other's
//     MFMAs; one barrier per block ("the next block has landed for every wave"), taken BEFORE the block's stores;
//   * a 32-column block is exactly one head: query blocks leave as fp32 rows (D^T = W_blk . X^T: a lane owns one row, 4 x 16 bytes),
//     key blocks in the same orientation as 8-byte plane entries ([dim group][key][8]: the two halves of a key are adjacent), value
//     blocks with the operands SWAPPED (D = X . W_blk^T: a lane owns one dim and 4 x 4 consecutive keys = the V^T image's entries).
// Counted waits: per block and wave the kernel issues RS_PIECES DMA requests and RS_E(kind) stores, unconditionally except in a job
// that reaches past row M (the last job of its workgroup), which drains instead of counting.
constexpr int RS_BLK = NPL * 16 * 2 * 32 * 8;        // 16-bit elements of one weight block (32 columns x 256 k x NPL planes = 32 KB)
constexpr int RS_RING = 4;
constexpr int RS_PIECES = RS_BLK / (512 * 8);        // 16-byte-per-thread DMA pieces of a block (4)
constexpr int RS_MAXB = 3 * DM / 32;                 // column blocks of the largest launch (in_proj: 24)
#define RS_LDS_BYTES (RS_RING * RS_BLK * 2 + RS_MAXB * 32 * 4)
#define RS_Q_ORI 0
#define RS_AHEAD 2         // weight blocks are requested this many phases ahead (2 or 3 with the four ring slots; 3 measured 1 % slower)
#define RS_PF 2            // LDS fragment prefetch distance in k-steps
#define RS_ST(P, V) __builtin_nontemporal_store(V, P)     // 512-byte runs that nobody re-reads before the attention kernel: -4 %
#define RS_STAMP(i)
__global__ __launch_bounds__(512, 2) void inproj_rs_kernel(const float* __restrict__ A, int lda, const op_t* __restrict__ Wb,
                                                           const float* __restrict__ bias, float* __restrict__ C, int ldc, int M,
                                                           int nb, const KvImg kv) {
  static_assert(NPL == 2 && RS_PIECES == 4, "the counted vmcnt waits below assume 4 DMA pieces per block and 16 (4) / 8 / 8 stores per epilogue");
  extern __shared__ __attribute__((aligned(16))) op_t rs_ring[];
  float* const bs = reinterpret_cast<float*>(rs_ring + RS_RING * RS_BLK);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  // the work is the sequence of (256-row job, 32-column block) phases, job-major; every workgroup takes an equal contiguous share of it
  // (a share may begin and end inside a job: its rows are then loaded by two workgroups)
  const int n_rb = (M + 255) / 256;
  const long total = (long)n_rb * nb;
  const int p_begin = (int)(total * blockIdx.x / gridDim.x), p_end = (int)(total * (blockIdx.x + 1) / gridDim.x);
  if (p_begin >= p_end) return;
  for (int i = tid; i < nb * 32; i += 512) bs[i] = bias ? bias[i] : 0.f;
  auto dma_piece = [&](int blk, int slot, int j) {
    const op_t* src = Wb + (size_t)blk * RS_BLK + (j * 512 + tid) * 8;
    op_t* dst = rs_ring + slot * RS_BLK + (j * 512 + wave * 64) * 8;       // wave-uniform LDS base (+ 16 B per lane)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  int job = p_begin / nb, cb = p_begin - job * nb;
  {
    const int c1 = cb + 1 == nb ? 0 : cb + 1;
#pragma unroll
    for (int j = 0; j < RS_PIECES; ++j) dma_piece(cb, 0, j);
#pragma unroll
    for (int j = 0; j < RS_PIECES; ++j) dma_piece(c1, 1, j);
  }
  int nxt = (cb + RS_AHEAD) % nb;                     // column block RS_AHEAD phases ahead
  const int kb0 = kv.k_col0 >> 5;                     // first key block; values from kb0 + 8
  constexpr int KIMG = 2 * NPL * 64 * HD, KPL = 64 * HD;
  int slot = 0;                                       // ring slot of the current block
  // The wave's 32 rows of a job arrive as 32 raw 16-byte loads per lane (k-step ks: k = 16 ks + 8 half .. + 7), all requested at the top
  // of the job and converted in order as they land.  (Requested a block earlier — right after the previous job's last k-loop, when the
  // fragment registers are dead — the job start shrinks from 20 % to 3 % of the wave's time and the barrier waits grow by the same
  // amount: the kernel is bound by what it writes, not by latencies.  profiles/README.md, round 4.)
  f32x4 raw[32];
  auto issue_rows = [&](int job_) {
    const int r_ = job_ * 256 + wave * 32 + l31;
    const float* xp = A + (size_t)(r_ < M ? r_ : M - 1) * lda + half * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      raw[2 * ks] = *reinterpret_cast<const f32x4*>(xp + ks * 16);         // (plain loads: a 128-byte line is fetched by four of them)
      raw[2 * ks + 1] = *reinterpret_cast<const f32x4*>(xp + ks * 16 + 4);
    }
  };
  int p = p_begin;
  while (p < p_end) {
    issue_rows(job);
    RS_STAMP(7)
    const int cbm = __builtin_amdgcn_readfirstlane(job * 256 + wave * 32);
    const int row = cbm + l31;
    const bool partial = job * 256 + 256 > M;         // workgroup-uniform: some lane of this job issues no stores
    // where this lane's row (key blocks) and its four row quads (value blocks) go in the images of head 0: resolved ONCE per job, before
    // the operand fragments are live (the class table stays out of the block loop); tile index < 0 = beyond M, nothing to store
    int k_tile = -1, k_meta = 0;                      // tile index; key in the tile (value blocks: key quad) | tiles per head << 6
    int v_tile[4], v_meta[4];
    {
      const KvTile kt_ = kv_tile(kv, cbm, 32);
      int b, pos, nkt;
      long tile0;
      if (row < M && kv.img) {
        kv_place(kv, kt_, cbm, row, b, pos, nkt, tile0);
        k_tile = (int)tile0 + b * NHEAD * nkt + (pos >> 6); k_meta = (pos & 63) | (nkt << 6);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int grow0 = cbm + 8 * g + 4 * half;                          // rows grow0 .. grow0 + 3: L % 4 == Lreg % 4 == 0, a quad never straddles
        v_tile[g] = -1; v_meta[g] = 0;
        if (grow0 < M && kv.img) {
          kv_place(kv, kt_, cbm, grow0, b, pos, nkt, tile0);
          v_tile[g] = (int)tile0 + b * NHEAD * nkt + (pos >> 6); v_meta[g] = ((pos & 63) >> 2) | (nkt << 6);
        }
      }
    }
    opx8 xT[16][NPL];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const f32x4 x0 = raw[2 * ks], x1 = raw[2 * ks + 1];
      const float xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      split_frag(xs, xT[ks]);
    }
    RS_STAMP(0)
    // (the fragment loads above were waited for with everything older complete: the next block's pieces have landed for this wave)
    if (p == p_begin) __syncthreads();                // blocks 0 and 1 of the share are in LDS, the bias vector is visible
    int e_prev = 0, e_prev2 = 0;                      // stores this wave issued in the previous phase / the one before (0: waiting for the rows above drained them)
    const int cb_last = (p_end - p) < (nb - cb) ? cb + (p_end - p) - 1 : nb - 1;   // last block of this job in the share
    // one quarter (quad q) of a finished block: fp32 row pieces / key plane entries / value plane entries; 1 / 2 / 2 stores
    auto epi_quad = [&](const f32x16& v, int knd, int cbv, int q) {
      const f32x4 x = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]} * WSCALE_INV;
      if (knd == 0) {
        if (row < M) *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + cbv * 32 + 4 * half + 8 * q) = x;
      } else if (knd == 1) {
        if (k_tile >= 0) {
          u32x2 pa[NPL];
          split_quad(x, pa);
          op_t* dst = kv.img + (size_t)(k_tile + (cbv - kb0) * (k_meta >> 6)) * KIMG + (k_meta & 63) * 8 + 4 * half + q * 64 * 8;
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) RS_ST(reinterpret_cast<u32x2*>(dst + pl * KPL), pa[pl]);
        }
      } else {
        if (v_tile[q] >= 0) {
          u32x2 pv[NPL];
          split_quad(x, pv);
          op_t* dst = kv.img + (size_t)(v_tile[q] + (cbv - kb0 - NHEAD) * (v_meta[q] >> 6)) * KIMG + NPL * KPL + ((v_meta[q] & 63) * HD + l31) * 4;
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) RS_ST(reinterpret_cast<u32x2*>(dst + pl * KPL), pv[pl]);
        }
      }
    };
    for (; cb <= cb_last; ++cb, ++p) {
      const int kind = cb < kb0 ? 0 : (cb < kb0 + NHEAD ? 1 : 2);          // 0 = fp32 rows, 1 = keys, 2 = values
      const op_t* w1 = rs_ring + slot * RS_BLK + (half * 32 + l31) * 8;    // [p][ks][half][col][8]
      const int nslot = (slot + RS_AHEAD) & 3;
      f32x16 acc;
      if (kind == 1 || (kind == 0 && RS_Q_ORI == 0)) {
        const float* bp = bs + cb * 32 + 4 * half;                         // register r <-> column (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + 8 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[4 * g + j] = bv[j] * WSCALE;
        }
      } else {
        const float bv = bs[cb * 32 + l31] * WSCALE;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bv;
      }
      // 48 MFMAs, the DMA pieces of the block two phases ahead at k-steps 0-3
      // 48 MFMAs in one chain (two chains, and the two waves of a SIMD taking their epilogues at different ends of the barrier, measured
      // no different: profiles/README.md), the DMA pieces of the block two phases ahead at k-steps 0-3
      auto kloop = [&](auto ORI) {
        constexpr int ori = decltype(ORI)::value;
        opx8 wf[RS_PF + 1][NPL];
        auto ld1 = [&](int ks, opx8 (&f)[NPL]) {
#pragma unroll
          for (int pp = 0; pp < NPL; ++pp) f[pp] = *reinterpret_cast<const opx8*>(w1 + ((pp * 16 + ks) * 2) * 32 * 8);
        };
#pragma unroll
        for (int i = 0; i < RS_PF; ++i) ld1(i, wf[i]);
        __builtin_amdgcn_sched_barrier(0);             // (keeps the fragment reads RS_PF k-steps ahead instead of all at the top: 256 registers)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (ks + RS_PF < 16) ld1(ks + RS_PF, wf[(ks + RS_PF) % (RS_PF + 1)]);
          if (ks < RS_PIECES) dma_piece(nxt, nslot, ks);
          const int c = ks % (RS_PF + 1);
          if (ori == 0) { SPLIT_TERMS(acc, wf[c], xT[ks]) }                // D^T = W_blk . X^T: a lane owns one row
          else { SPLIT_TERMS(acc, xT[ks], wf[c]) }                         // D = X . W_blk^T: a lane owns one dim
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      RS_STAMP(4)
      if (kind == 2 || (kind == 0 && RS_Q_ORI == 1)) kloop(std::integral_constant<int, 1>{});
      else kloop(std::integral_constant<int, 0>{});
      RS_STAMP(1)
      const int e_cur = kind == 0 ? (RS_Q_ORI ? 16 : 4) : 8;
      // the block of the NEXT phase (requested in the previous phase's k-steps 0-3) must have landed; younger requests of this wave, in
      // issue order: the previous phase's stores, then this phase's RS_PIECES pieces.  vmcnt counts in order; stores count.
      {
        // RS_AHEAD == 3: the awaited block was requested TWO phases ago — behind it in the queue: that phase's stores, the last phase's pieces
        // and stores, this phase's pieces; what must have drained by now was issued three phases ago, not two
        const int younger = partial ? 0 : (RS_AHEAD == 3 ? e_prev2 + e_prev + 2 * RS_PIECES : e_prev + RS_PIECES);
        switch (younger) {                                                             // vmcnt: bits 3:0 and 15:14
          case 0: __builtin_amdgcn_s_waitcnt(0x0070); break;
          case 4: __builtin_amdgcn_s_waitcnt(0x0070 | 4); break;
          case 8: __builtin_amdgcn_s_waitcnt(0x0070 | 8); break;
          case 12: __builtin_amdgcn_s_waitcnt(0x0070 | 12); break;
          case 16: __builtin_amdgcn_s_waitcnt(0x4070 | 0); break;
          case 20: __builtin_amdgcn_s_waitcnt(0x4070 | 4); break;
          case 24: __builtin_amdgcn_s_waitcnt(0x4070 | 8); break;
          case 28: __builtin_amdgcn_s_waitcnt(0x4070 | 12); break;
          case 32: __builtin_amdgcn_s_waitcnt(0x8070 | 0); break;
          case 36: __builtin_amdgcn_s_waitcnt(0x8070 | 4); break;
          default: __builtin_amdgcn_s_waitcnt(0x8070 | 8); break;                      // 40 = 16 + 16 + 8 (query-line variant)
        }
      }
      __builtin_amdgcn_s_barrier();                   // taken BEFORE this block's stores: they leave underneath the next block's MFMAs
      RS_STAMP(2)
#pragma unroll
      for (int q = 0; q < 4; ++q) epi_quad(acc, kind, cb, q);
      RS_STAMP(3)
      e_prev2 = e_prev;
      e_prev = e_cur;
      slot = (slot + 1) & 3;
      nxt = nxt + 1 == nb ? 0 : nxt + 1;
    }
    if (cb == nb) cb = 0;
    ++job;
  }
}

int launch_inproj_rs(const float* A, int lda, const void* Wblk, const float* bias, float* C, int ldc, int M, int N, void* kv_img,
                     int kv_col0, int kv_n, const KvClassHost* kv_cls, hipStream_t st) {
  if (M <= 0) return CTRLSIM_OK;
  if (!A || !Wblk || (lda & 3) || (ldc & 3) || (N & 31) || N < 64 || N > 32 * RS_MAXB) return CTRLSIM_EINVAL;
  if (kv_img ? (!kv_cls || (kv_col0 & 31) || N != kv_col0 + 2 * DM || (kv_col0 && !C) || kv_n < 1 || kv_n > MAXC) : (!C || kv_col0 != N))
    return CTRLSIM_EINVAL;                               // kv_img == NULL: a plain Linear, every block leaves as fp32 rows
  KvImg kv;
  kv.img = static_cast<op_t*>(kv_img); kv.k_col0 = kv_col0; kv.n = 0;
  int row0 = kv_img ? 0 : M;
  kv.c[0] = KvClass{0, 32, 32, 0, 1, 0};                 // (read, never used, by the kernel's placement code when there are no images)
  for (int k = 0; kv_img && k < kv_n; ++k) {
    const KvClassHost& c = kv_cls[k];
    if (c.B <= 0) continue;
    if ((c.L & 3) || c.L < 32 || (c.Lreg & 3) || c.Lreg > c.L || c.Lreg <= 0 || (c.rep_k0 & 63) ||
        (c.Lreg < c.L && c.rep_k0 < c.Lreg) || c.nkt * 64 < (c.Lreg < c.L ? c.rep_k0 + (c.L - c.Lreg) : c.L))
      return CTRLSIM_EINVAL;
    kv.c[kv.n++] = KvClass{row0, c.L, c.Lreg, c.rep_k0, c.nkt, c.tile0};
    row0 += c.B * c.L;
  }
  if (row0 != M) return CTRLSIM_EINVAL;
  static const int cus = [] {
    int dev = 0, n = 0;
    return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }();
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&inproj_rs_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS_BYTES) == hipSuccess;
  if (!attr_ok) return CTRLSIM_EINVAL;
  const int n_rb = (M + 255) / 256;
  prof_before(PROF_GEMM, st);
  const long shares = (long)n_rb * (N / 32) / 4;         // at least four (job, block) phases per workgroup
  hipLaunchKernelGGL(inproj_rs_kernel, dim3(shares < 1 ? 1 : (shares < cus ? (int)shares : cus)), dim3(512), RS_LDS_BYTES, st, A, lda,
                     static_cast<const op_t*>(Wblk), bias, C, ldc, M, N / 32, kv);
  const double MN = (double)M * N, kvN = kv_img ? 2.0 * DM : 0.0;
  prof_after(PROF_GEMM, 2.0 * MN * (double)DM, st,
             4.0 * (double)M * DM + 4.0 * (double)M * (N - kvN) + 2.0 * NPL * (double)M * kvN + 2.0 * NPL * (double)N * DM,
             kv_img ? PKIND_GEMM_QKV_KV : PKIND_GEMM_PLAIN);
  return ctrlsim_launch_status();
}
#else
int launch_inproj_rs(const float*, int, const void*, const float*, float*, int, int, int, void*, int, int, const KvClassHost*, hipStream_t) {
  return CTRLSIM_EINVAL;                               // two-fp16-plane scheme only
}
#endif

// Plain Linear 256 -> 256 whose result row i is written to row c_rows[i] of C (entries < 0 are not stored): the weight-stationary kernel
// with a scattered row store.  Returns 1 (nothing launched) when that kernel does not apply — other split scheme, option off, other
// shape — and the caller runs Linear + row copy instead.
int launch_gemm256_rows(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, float* C, int ldc,
                        const int* c_rows, int M, hipStream_t st) {
  if (M <= 0) return CTRLSIM_OK;
#if CTRLSIM_F16X3
  if (!A || !W3 || !C || !c_rows || (lda & 3) || (ldc & 3) || n0 < 0 || n0 + 256 > n_total) return CTRLSIM_EINVAL;
  if (!(ctrlsim_option(OPT_GEMM_WS) & (M >= 2 * WS_ROWS * 256 ? 1 : 2))) return 1;
  const int nblk = (M + WS_ROWS - 1) / WS_ROWS;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws256_kernel<false, false, false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES) == hipSuccess;
  if (!attr_ok) return CTRLSIM_EINVAL;
  KvImg kv;
  kv.img = nullptr; kv.k_col0 = 0; kv.n = 0;
  prof_before(PROF_GEMM, st);
  hipLaunchKernelGGL((gemm_ws256_kernel<false, false, false>), dim3(nblk < cus ? nblk : cus), dim3(512), WS_LDS_BYTES, st, A, lda,
                     static_cast<const op_t*>(W3), bias, nullptr, nullptr, nullptr, 0, C, ldc, M, n_total, n0, ctrlsim_nonfinite_ptr(), kv,
                     c_rows);
  prof_after(PROF_GEMM, 2.0 * (double)M * 256.0 * 256.0, st, 8.0 * (double)M * 256.0 + 2.0 * NPL * 256.0 * 256.0, PKIND_GEMM_PLAIN);
  return ctrlsim_launch_status();
#else
  return 1;
#endif
}

int launch_gemm_nt_bf16x6_kvc(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias,
                              const float* R, int ldr, float* C, int ldc, int M, int N, int K, int relu,
                              const float* ln_gamma, const float* ln_beta, void* kv_img, int kv_col0, int kv_n,
                              const KvClassHost* kv_cls, hipStream_t st) {
  if (M <= 0) return CTRLSIM_OK;
  KvImg kv;
  kv.img = static_cast<op_t*>(kv_img); kv.k_col0 = kv_col0; kv.n = 0;
  if (kv_img) {
    if (ln_gamma || R || relu || (kv_col0 & 127) || N != kv_col0 + 2 * DM || kv_n < 1 || kv_n > MAXC || !kv_cls) return CTRLSIM_EINVAL;
    int row0 = 0;
    for (int k = 0; k < kv_n; ++k) {
      const KvClassHost& c = kv_cls[k];
      if (c.B <= 0) continue;
      if ((c.L & 3) || c.L < 32 || (c.Lreg & 3) || c.Lreg > c.L || c.Lreg <= 0 || (c.rep_k0 & 63) ||
          (c.Lreg < c.L && c.rep_k0 < c.Lreg) || c.nkt * 64 < (c.Lreg < c.L ? c.rep_k0 + (c.L - c.Lreg) : c.L))
        return CTRLSIM_EINVAL;
      kv.c[kv.n++] = KvClass{row0, c.L, c.Lreg, c.rep_k0, c.nkt, c.tile0};
      row0 += c.B * c.L;
    }
    if (row0 != M) return CTRLSIM_EINVAL;
  }
  if (K % XK != 0 || (lda & 3) || N <= 0 || !W3 || n0 < 0 || n0 + N > n_total) return CTRLSIM_EINVAL;
  const bool ln = ln_gamma != nullptr;
  if (ln && (N != 256 || (ldc & 3) || (R && (ldr & 3)))) return CTRLSIM_EINVAL;
  if (!ln && R && relu) return CTRLSIM_EINVAL;
#if CTRLSIM_F16X3
  // OPT_GEMM_WS: bit 0 = launches of at least two row blocks per CU, bit 1 = the smaller ones, bit 2 = the K / V-image Linears
  const bool ws_kv = kv_img && K == 256 && !(N & 255) && !(kv_col0 & 255) && !(ldc & 3) && (ctrlsim_option(OPT_GEMM_WS) & 4);
  if (ws_kv || (!kv_img && N == 256 && K == 256 && !(ldc & 3) && (!R || !(ldr & 3)) &&
                (ctrlsim_option(OPT_GEMM_WS) & (M >= 2 * WS_ROWS * 256 ? 1 : 2)))) {
    const int nblk = (M + WS_ROWS - 1) / WS_ROWS;
    static const int cus = [] {
      int dev = 0, n = 0;
      return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }();
    dim3 g(nblk < cus ? nblk : cus, ws_kv ? N / 256 : 1), b(512);
    const op_t* w = static_cast<const op_t*>(W3);
    int* nonfinite = ctrlsim_nonfinite_ptr();
    prof_before(PROF_GEMM, st);
    if (ws_kv) {
      static const bool attr_ok =
          hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws256_kernel<false, false, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES) == hipSuccess;
      if (!attr_ok) return CTRLSIM_EINVAL;
      hipLaunchKernelGGL((gemm_ws256_kernel<false, false, false, true>), g, b, WS_LDS_BYTES, st, A, lda, w, bias, nullptr, nullptr,
                         nullptr, 0, C, ldc, M, n_total, n0, nonfinite, kv);
      const double MN = (double)M * N, kvN = 2.0 * DM;
      prof_after(PROF_GEMM, 2.0 * MN * (double)K, st,
                 4.0 * (double)M * K + 4.0 * (double)M * (N - kvN) + 2.0 * NPL * (double)M * kvN + 2.0 * NPL * (double)N * K,
                 PKIND_GEMM_QKV_KV);
      return ctrlsim_launch_status();
    }
#define WS_LAUNCH(RELU_, RESID_, LN_)                                                                                  \
  do {                                                                                                                 \
    static const bool attr_ok =                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws256_kernel<RELU_, RESID_, LN_>),                     \
                            hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES) == hipSuccess;                   \
    if (!attr_ok) return CTRLSIM_EINVAL;                                                                               \
    hipLaunchKernelGGL((gemm_ws256_kernel<RELU_, RESID_, LN_>), g, b, WS_LDS_BYTES, st, A, lda, w, bias, ln_gamma,     \
                       ln_beta, R, ldr, C, ldc, M, n_total, n0, nonfinite, kv);                                        \
  } while (0)
    if (ln) {
      if (R && relu) WS_LAUNCH(true, true, true);
      else if (R) WS_LAUNCH(false, true, true);
      else if (relu) WS_LAUNCH(true, false, true);
      else WS_LAUNCH(false, false, true);
    } else if (R) WS_LAUNCH(false, true, false);
    else if (relu) WS_LAUNCH(true, false, false);
    else WS_LAUNCH(false, false, false);
#undef WS_LAUNCH
    const double MN = (double)M * N;
    prof_after(PROF_GEMM, 2.0 * MN * (double)K, st, 4.0 * (double)M * K + 4.0 * MN + (R ? 4.0 * MN : 0.0) + 2.0 * NPL * (double)N * K,
               ln ? PKIND_GEMM_LN : PKIND_GEMM_PLAIN);
    return ctrlsim_launch_status();
  }
#endif
  const int tile_opt = ctrlsim_option(OPT_GEMM6_TILE);          // 0 = auto, 1 = 128x128, 2 = 64x256 (A/B knob)
  const bool wide = !kv_img && (ln || tile_opt == 2);
  const bool small = !wide && tile_opt == 3;                     // 64x128 tiles (wave tile 32x64), 4 workgroups per CU
  const int XM = (wide || small) ? 64 : 128, XN = wide ? 256 : 128;
  const int m_tiles = (M + XM - 1) / XM, n_tiles = (N + XN - 1) / XN;
  const int total = ((m_tiles + 7) / 8) * 8 * n_tiles;
  const int resident = 256 * (wide ? GEMM_OCC_14 : (small ? 4 : GEMM_OCC_22));
  const int grid = total < resident ? total : resident;
  dim3 g(grid), b(256);
  const op_t* w = static_cast<const op_t*>(W3);
  // the k-step stages (NPL planes each), or the epilogue's fp32 chunk (32 rows per wave row x XN + 4 columns), whichever is larger
  const size_t shm_k = (size_t)(wide ? GEMM_RING(1, 4) : GEMM_RING(2, 2)) * NPL * (XM + XN) * XK * sizeof(op_t);
  const size_t shm_e = (size_t)(wide ? 32 : 64) * (XN + 4) * sizeof(float);
  const size_t shm = shm_k > shm_e ? shm_k : shm_e;
#define GEMM6_LAUNCH(WR_, WC_, RELU_, RESID_, LN_, KV_) GEMM6_LAUNCH_(WR_, WC_, 2, RELU_, RESID_, LN_, KV_)
#define GEMM6_LAUNCH_(WR_, WC_, MR_, RELU_, RESID_, LN_, KV_)                                                               \
  do {                                                                                                                \
    static const bool attr_ok =      /* once per instantiation, thread-safe static initialisation */                  \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16x6_kernel<WR_, WC_, MR_, RELU_, RESID_, LN_, KV_>), \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess;                      \
    if (!attr_ok) return CTRLSIM_EINVAL;                                                                              \
    hipLaunchKernelGGL((gemm_nt_bf16x6_kernel<WR_, WC_, MR_, RELU_, RESID_, LN_, KV_>), g, b, shm, st, A, lda, w, bias,    \
                       ln_gamma, ln_beta, R, ldr, C, ldc, M, N, K, m_tiles, n_tiles, n_total, n0, kv, nonfinite);                \
  } while (0)
  int* nonfinite = ctrlsim_nonfinite_ptr();
  prof_before(PROF_GEMM, st);
  if (kv_img) {
    if (small) GEMM6_LAUNCH_(2, 2, 1, false, false, false, true);
    else GEMM6_LAUNCH(2, 2, false, false, false, true);
  } else if (small) {
    if (R) GEMM6_LAUNCH_(2, 2, 1, false, true, false, false);
    else if (relu) GEMM6_LAUNCH_(2, 2, 1, true, false, false, false);
    else GEMM6_LAUNCH_(2, 2, 1, false, false, false, false);
  } else if (ln) {
    if (R && relu) GEMM6_LAUNCH(1, 4, true, true, true, false);
    else if (R) GEMM6_LAUNCH(1, 4, false, true, true, false);
    else if (relu) GEMM6_LAUNCH(1, 4, true, false, true, false);
    else GEMM6_LAUNCH(1, 4, false, false, true, false);
  } else if (wide) {
    if (R) GEMM6_LAUNCH(1, 4, false, true, false, false);
    else if (relu) GEMM6_LAUNCH(1, 4, true, false, false, false);
    else GEMM6_LAUNCH(1, 4, false, false, false, false);
  } else {
    if (R) GEMM6_LAUNCH(2, 2, false, true, false, false);
    else if (relu) GEMM6_LAUNCH(2, 2, true, false, false, false);
    else GEMM6_LAUNCH(2, 2, false, false, false, false);
  }
#undef GEMM6_LAUNCH
#undef GEMM6_LAUNCH_
  {
    const double MN = (double)M * N, kvN = kv_img ? 2.0 * DM : 0.0;
    const double out_bytes = 4.0 * (double)M * (N - kvN) + 2.0 * NPL * (double)M * kvN;   // K / V columns leave as NPL 16-bit planes
    prof_after(PROF_GEMM, 2.0 * MN * (double)K, st,
               4.0 * (double)M * K + out_bytes + (R ? 4.0 * MN : 0.0) + 2.0 * NPL * (double)N * K,   // weights: NPL 16-bit planes
               kv_img ? PKIND_GEMM_QKV_KV : ln ? PKIND_GEMM_LN : PKIND_GEMM_PLAIN);
  }
  return ctrlsim_launch_status();
}

}  // namespace SPLIT_NS
