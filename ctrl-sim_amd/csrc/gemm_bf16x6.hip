// fp32-accurate GEMM on the bf16 MFMA: every fp32 operand is split into three bf16 terms (x = hi + mid + lo, 3 x 8
// significant bits = the whole fp32 mantissa) and the product is evaluated as the six leading partial products
//     a.b ~= hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid          (dropped terms <= 2^-23 relative)
// with fp32 accumulation inside v_mfma_f32_32x32x16_bf16.  The result carries fp32-class error (measured against an
// fp64 reference in tests/test_gpu_ops.py: same order as the f32-input MFMA kernel) while the matrix pipe runs at
// 16x the f32-input MFMA rate for 6x the instructions: a 2.67x higher matrix roof (420 TFLOP/s fp32-equivalent).
// Token parity with the reference is unaffected (fixtures: bit-exact tokens, logits within 1e-4).
//
// Weights are split ONCE at pack time (ctrlsim_amd/pack.py) into slab-major planes  W3[K/16][3][2][N][8] bf16  (the
// two 8-element halves of a 16-wide k-step are separate sub-planes) so a workgroup's K-slab of a sub-plane is one
// contiguous 16-byte-per-row stream and the LDS image [plane][half][row][8] makes every fragment read of a wave two
// contiguous 512-byte spans (no bank conflicts; rows adjacent in a 32-byte layout were 2-way conflicting).  Activations stay fp32 in HBM and are split
// in registers while being staged into LDS (v_cvt_pk_bf16_f32, round-to-nearest-even).
//
// Tiling: workgroup = 128 (M) x 256 (N), 4 waves as 2x2, wave tile 64x128 = 2x4 MFMA tiles (128 accumulator regs);
// extending N (pre-split weights) rather than M amortises the activation split.  K-slab = 16 = one MFMA k-step;
// LDS per buffer: A planes 3 x [128][16] + W planes 3 x [256][16] bf16 = 36 KB, double buffered (72 KB, 2 WG / CU),
// rows are 32 bytes so a wave's 16-byte fragment reads are one contiguous 1-2 KB span (conflict free, no padding).
// Persistent workgroups with the XCD-aware tile order of gemm.hip; the next tile's first slab is prefetched before
// the epilogue; the epilogue stages 64 rows at a time through LDS for 16-byte bias / residual / store traffic.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define XM 128
#define XN 256
#define XK 16

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// one v_cvt_pk_bf16_f32 converts AND packs two values (RNE); 11 VALU ops per pair for the three planes
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xFFFF0000u);
  mid = cvt_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xFFFF0000u);
  lo = cvt_pk_bf16(sa, sb);
}
__device__ __forceinline__ void split3(const f32x4 x, u32x2& hi, u32x2& mid, u32x2& lo) {
  unsigned h0, m0, l0, h1, m1, l1;
  split3_pair(x[0], x[1], h0, m0, l0);
  split3_pair(x[2], x[3], h1, m1, l1);
  hi = u32x2{h0, h1}; mid = u32x2{m0, m1}; lo = u32x2{l0, l1};
}

template <int PA, int PB>
__device__ __forceinline__ void term(f32x16 (&acc)[2][2], const bf16x8 (&fa)[2][3], const bf16x8 (&fb)[2][3]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][PA], fb[b][PB], acc[a][b], 0, 0, 0);
}

// 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 64 x 64 (2x2 MFMA tiles); one LDS stage = 32 k (two 16-wide k-steps),
// two stages (144 KB, one workgroup per CU, two waves per SIMD), one barrier per 32 k.
template <bool RELU, bool RESID>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16x6_kernel(const float* __restrict__ A, int lda,
                                                                const __bf16* __restrict__ W3,   // [K/16][3][2][N][8]
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ R, int ldr,
                                                                float* __restrict__ C, int ldc, int M, int N, int K,
                                                                int m_tiles, int n_tiles, int n_total, int n0) {
  // W3 holds all n_total rows of the packed matrix; this GEMM uses rows [n0, n0 + N) (e.g. the q / kv halves of an
  // in_proj_weight)
  constexpr int A_PLANE = XM * XK;                 // bf16 elements of one plane of one k-step
  constexpr int W_PLANE = XN * XK;
  constexpr int KSBUF = 3 * A_PLANE + 3 * W_PLANE; // one k-step (18432 bf16 = 36 KB)
  constexpr int BUF = 2 * KSBUF;                   // one stage = two k-steps (72 KB)
  constexpr int CP = XN + 4;
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];   // 2 * BUF bf16 = 144 KB
  static_assert(2 * BUF * 2 >= 64 * CP * 4, "epilogue staging must fit");

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;

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
  // the W part of a stage is lane-linear in exactly the order idx = tid + 512*i, so the DMA needs no VGPRs at all)
  f32x4 ra[2];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const size_t w_slab = (size_t)3 * n_total * XK;  // bf16 elements per 16-wide K-slab of W3
  auto gload_a = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, r = idx >> 3, c = (idx & 7) * 4;
      const int ga = bm + r;
      ra[i] = ga < M ? *reinterpret_cast<const f32x4*>(A + (size_t)ga * lda + kt * 32 + c) : zero4;
    }
  };
  auto dma_w = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = tid + 512 * i, ks = idx / 1536, rem = idx - ks * 1536, pc = rem >> 8, r = rem & 255;
      int gw = bn + r;
      gw = gw < N ? gw : N - 1;                    // columns >= N are computed on clamped rows and never stored
      const __bf16* src = W3 + (size_t)(2 * kt + ks) * w_slab + ((size_t)pc * n_total + n0 + gw) * 8;
      const int idx0 = wave * 64 + 512 * i, ks0 = idx0 / 1536, rem0 = idx0 - ks0 * 1536;   // wave-uniform LDS base
      __bf16* dst = lds + buf * BUF + ks0 * KSBUF + 3 * A_PLANE + rem0 * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  auto sstore_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, r = idx >> 3, c = (idx & 7) * 4;
      u32x2 hi, mid, lo;
      split3(ra[i], hi, mid, lo);
      __bf16* Ab = lds + buf * BUF + (c >> 4) * KSBUF + ((c >> 3) & 1) * (A_PLANE / 2) + r * 8 + (c & 7);
      *reinterpret_cast<u32x2*>(Ab + 0 * A_PLANE) = hi;
      *reinterpret_cast<u32x2*>(Ab + 1 * A_PLANE) = mid;
      *reinterpret_cast<u32x2*>(Ab + 2 * A_PLANE) = lo;
    }
  };

  const int nk = K / 32;
  int id = blockIdx.x;
  while (id < total_ids && !tile_of(id, bm, bn)) id += gridDim.x;
  if (id >= total_ids) return;
  gload_a(0);
  for (;;) {
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    dma_w(0, 0);
    sstore_a(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) {
#ifndef ABL_NO_DMA
        dma_w(kt + 1, cur ^ 1);
#endif                    // stage cur^1 was released by the barrier that ended kt-1
        gload_a(kt + 1);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const __bf16* Ab = lds + cur * BUF + ks * KSBUF + half * (A_PLANE / 2) + (wr * 64 + l31) * 8;
        const __bf16* Wb = lds + cur * BUF + ks * KSBUF + 3 * A_PLANE + half * (W_PLANE / 2) + (wc * 64 + l31) * 8;
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int a = 0; a < 2; ++a) fa[a][p] = *reinterpret_cast<const bf16x8*>(Ab + p * A_PLANE + a * 32 * 8);
#pragma unroll
          for (int b = 0; b < 2; ++b) fb[b][p] = *reinterpret_cast<const bf16x8*>(Wb + p * W_PLANE + b * 32 * 8);
        }
        // six partial products, smallest first; term-major order keeps 4 independent accumulators between reuses
#ifndef ABL_NO_MFMA
        term<2, 0>(acc, fa, fb);
        term<0, 2>(acc, fa, fb);
        term<1, 1>(acc, fa, fb);
        term<1, 0>(acc, fa, fb);
        term<0, 1>(acc, fa, fb);
        term<0, 0>(acc, fa, fb);
#else
#pragma unroll
        for (int p = 0; p < 3; ++p) { asm volatile("" ::"v"(fa[0][p]), "v"(fa[1][p]), "v"(fb[0][p]), "v"(fb[1][p])); }
#endif
      }
      if (kt + 1 < nk) sstore_a(cur ^ 1);
      __syncthreads();
    }

    // ---- epilogue: two chunks of 64 rows (tile row a of both wave rows) staged through LDS
    float* Cs = reinterpret_cast<float*>(lds);   // [64][XN + 4]
    const bool vec_ok = !(ldc & 3) && (!RESID || !(ldr & 3));
    const int cbm = bm, cbn = bn;
    int nid = id + gridDim.x;
    while (nid < total_ids && !tile_of(nid, bm, bn)) nid += gridDim.x;
    const bool have_next = nid < total_ids;
    if (have_next) gload_a(0);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[(wr * 32 + mfma_row(r, half)) * CP + wc * 64 + b * 32 + l31] = acc[a][b][r];
      __syncthreads();
#pragma unroll 4
      for (int i = 0; i < 8; ++i) {
        const int idx = tid + 512 * i, lr = idx >> 6, col = (idx & 63) * 4;
        const int grow = cbm + (lr >> 5) * 64 + a * 32 + (lr & 31), gcol = cbn + col;
        if (grow >= M || gcol >= N) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(Cs + lr * CP + col);
        if (vec_ok && gcol + 3 < N) {
          if (bias) v += *reinterpret_cast<const f32x4*>(bias + gcol);
          if (RESID) v += *reinterpret_cast<const f32x4*>(R + (size_t)grow * ldr + gcol);
          if (RELU) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
          }
          *reinterpret_cast<f32x4*>(C + (size_t)grow * ldc + gcol) = v;
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

int launch_gemm_nt_bf16x6(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, const float* R, int ldr, float* C,
                          int ldc, int M, int N, int K, int relu, hipStream_t st) {
  if (M <= 0) return CTRLSIM_OK;
  if (K % 32 != 0 || (lda & 3) || N <= 0 || !W3 || n0 < 0 || n0 + N > n_total) return CTRLSIM_EINVAL;
  const int m_tiles = (M + XM - 1) / XM, n_tiles = (N + XN - 1) / XN;
  const int total = ((m_tiles + 7) / 8) * 8 * n_tiles;
  const int resident = 256;                        // one 512-thread workgroup per CU (144 KB of LDS)
  const int grid = total < resident ? total : resident;
  dim3 g(grid), b(512);
  const __bf16* w = static_cast<const __bf16*>(W3);
  const size_t shm = (size_t)2 * 2 * (3 * XM * XK + 3 * XN * XK) * sizeof(__bf16);   // 147456 B
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16x6_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16x6_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16x6_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_set = true;
  }
  prof_before(PROF_GEMM, st);
  if (R) {
    if (relu) return CTRLSIM_EINVAL;
    hipLaunchKernelGGL((gemm_nt_bf16x6_kernel<false, true>), g, b, shm, st, A, lda, w, bias, R, ldr, C, ldc, M, N, K, m_tiles,
                       n_tiles, n_total, n0);
  } else if (relu) {
    hipLaunchKernelGGL((gemm_nt_bf16x6_kernel<true, false>), g, b, shm, st, A, lda, w, bias, R, ldr, C, ldc, M, N, K, m_tiles,
                       n_tiles, n_total, n0);
  } else {
    hipLaunchKernelGGL((gemm_nt_bf16x6_kernel<false, false>), g, b, shm, st, A, lda, w, bias, R, ldr, C, ldc, M, N, K, m_tiles,
                       n_tiles, n_total, n0);
  }
  prof_after(PROF_GEMM, 2.0 * (double)M * (double)N * (double)K, st);
  return ctrlsim_launch_status();
}
