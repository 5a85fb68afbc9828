// fp32 MFMA GEMM + LayerNorm + small-K input layers for the CtRL-Sim transformer (gfx950).
//
// Every Linear of the reference model (torch.nn.Linear: y = x W^T + b, W row-major [N,K]; utils/layers.py:6-19,
// nn.TransformerEncoder/DecoderLayer in_proj/out_proj/linear1/linear2, modules/encoder.py:21-46) is this
// "NT" product with K-contiguous operands.  fp32 operands are forced by token parity, so the
// matrix instruction is v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 64 FLOP/clk/SIMD, 157 TF chip peak).
//
// Tiling (64-lane waves): block = 128x128 outputs, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of 32x32
// (64 accumulator VGPRs).  K is staged through LDS in 32-wide slabs, double buffered, rows padded to 36 floats
// so the per-lane 16-byte fragment reads (row = lane&31, k-offset = 4*(lane>>5)) are bank-conflict free.
// One ds_read_b128 per operand feeds 4 MFMAs: within an 8-wide k-chunk lanes 0-31 hold k=0..3 and lanes 32-63
// hold k=4..7 of their row; MFMA j consumes element j of both halves, i.e. the products (k=j, k=4+j) — the k
// labelling inside a chunk is free as long as A and B use the same one.
// Block ids are remapped so that all N-tiles of an M-tile run on one XCD (block b -> XCD b%8): the activation
// slab is then fetched into a single XCD's L2.
#include "common.h"

#define BM 128
#define BN 128

// TBK = K-slab width staged in LDS (16 or 32).  LDS = max(K-loop double buffer, epilogue staging of 64 rows):
//   TBK=32: 73.7 KB -> 2 workgroups / CU;   TBK=16: 41 KB -> 3 workgroups / CU (VGPR-limited), twice the barriers.
template <int TBK, bool RELU, bool RESID>
__global__ __launch_bounds__(256, (TBK == 16 ? 3 : 2)) void gemm_nt_f32_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias,
                                                          const float* R, int ldr,   // C may alias R: no restrict
                                                          float* C, int ldc, int M, int N, int K,
                                                          int m_tiles, int n_tiles) {
  constexpr int BKP = TBK + 4;
  constexpr int CP = BN + 4;
  constexpr int LOOP_FLOATS = 2 * (BM + BN) * BKP;
  constexpr int EPI_FLOATS = 64 * CP;
  __shared__ __attribute__((aligned(16))) float lds[LOOP_FLOATS > EPI_FLOATS ? LOOP_FLOATS : EPI_FLOATS];
  float* As = lds;
  float* Ws = lds + 2 * BM * BKP;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, half = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;

  // Persistent workgroups: tile ids id = blockIdx.x, += gridDim.x (gridDim.x is a multiple of 8, so a workgroup stays
  // on its XCD).  XCD-aware tile order: id -> (xcd = id & 7, j = id >> 3), M-tile = (j / n_tiles) * 8 + xcd,
  // N-tile = j % n_tiles: all N-tiles of an M-tile are consecutive on ONE XCD, the activation slab is fetched into a
  // single L2.  The first K-slab of the NEXT tile is loaded into registers before the epilogue of the current one.
  const int total_ids = ((m_tiles + 7) / 8) * 8 * n_tiles;
  int bm = 0, bn = 0;
  auto tile_of = [&](int id, int& tbm, int& tbn) -> bool {
    const int xcd = id & 7, j = id >> 3;
    const int mt = (j / n_tiles) * 8 + xcd, nt = j % n_tiles;
    tbm = mt * BM;
    tbn = nt * BN;
    return mt < m_tiles;
  };

  constexpr int F4_PER_ROW = TBK / 4;                 // float4 per tile row
  constexpr int LD_ITERS = (BM * F4_PER_ROW) / 256;   // float4 per thread and operand
  f32x4 ra[LD_ITERS], rw[LD_ITERS];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LD_ITERS; ++i) {
      const int idx = tid + 256 * i, r = idx / F4_PER_ROW, c = (idx % F4_PER_ROW) * 4;
      const int ga = bm + r, gw = bn + r;
      ra[i] = ga < M ? *reinterpret_cast<const f32x4*>(A + (size_t)ga * lda + k0 + c) : zero4;
      rw[i] = gw < N ? *reinterpret_cast<const f32x4*>(W + (size_t)gw * ldw + k0 + c) : zero4;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LD_ITERS; ++i) {
      const int idx = tid + 256 * i, r = idx / F4_PER_ROW, c = (idx % F4_PER_ROW) * 4;
      *reinterpret_cast<f32x4*>(As + buf * BM * BKP + r * BKP + c) = ra[i];
      *reinterpret_cast<f32x4*>(Ws + buf * BN * BKP + r * BKP + c) = rw[i];
    }
  };

  const int nk = K / TBK;
  int id = blockIdx.x;
  while (id < total_ids && !tile_of(id, bm, bn)) id += gridDim.x;
  if (id >= total_ids) return;
  gload(0);
  for (;;) {
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * TBK);
    const float* as = As + cur * BM * BKP + (wr * 64 + l31) * BKP + half * 4;
    const float* ws = Ws + cur * BN * BKP + (wc * 64 + l31) * BKP + half * 4;
#pragma unroll
    for (int kc = 0; kc < TBK / 8; ++kc) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(as + kc * 8);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(as + 32 * BKP + kc * 8);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(ws + kc * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(ws + 32 * BKP + kc * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b0[q], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b1[q], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b0[q], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b1[q], acc[1][1], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue: stage the tile through LDS in two 64-row halves (the K-loop buffers are free after its final barrier) so
  // that bias / residual loads and the stores are 16-byte, 512-byte-per-row transactions instead of dword accesses.
  // acc fragment: lane holds col = l31, rows (r&3)+8*(r>>2)+4*half of each 32x32 tile.
  float* Cs = lds;   // [64][BN + 4]
  const bool vec_ok = !(ldc & 3) && (!RESID || !(ldr & 3));
  const int cbm = bm, cbn = bn;            // this tile's origin (bm/bn move on to the prefetched tile)
  int nid = id + gridDim.x;
  while (nid < total_ids && !tile_of(nid, bm, bn)) nid += gridDim.x;
  const bool have_next = nid < total_ids;
  if (have_next) gload(0);                 // next tile's first slab: in flight during the epilogue
#pragma unroll
  for (int hrow = 0; hrow < 2; ++hrow) {
    if (wr == hrow) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            Cs[(a * 32 + mfma_row(r, half)) * CP + wc * 64 + b * 32 + l31] = acc[a][b][r];
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i, row = idx >> 5, col = (idx & 31) * 4;
      const int grow = cbm + hrow * 64 + row, gcol = cbn + col;
      if (grow >= M || gcol >= N) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * CP + col);
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

// y = LayerNorm(x [+ r]) * gamma + beta [ReLU], rows of 256; one wave per row (4 channels per lane).
// torch.nn.LayerNorm semantics: biased variance, eps inside the sqrt (eps = 1e-5 everywhere in the model).
template <bool RELU>
__global__ __launch_bounds__(256) void layernorm256_kernel(const float* X, int ldx,    // Y may alias X / Radd: no restrict
                                                           const float* Radd, int ldr,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* Y,
                                                           int ldy, int rows, int* __restrict__ nonfinite) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  f32x4 v = *reinterpret_cast<const f32x4*>(X + (size_t)row * ldx + lane * 4);
  if (Radd) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(Radd + (size_t)row * ldr + lane * 4);
    v += r;
  }
  const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
  const f32x4 d = v - mean;
  const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256.f);
  count_nonfinite_row(var, lane, nonfinite);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + lane * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(beta + lane * 4);
  f32x4 y = d * rstd * g + b;
  if (RELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = fmaxf(y[i], 0.f);
  }
  *reinterpret_cast<f32x4*>(Y + (size_t)row * ldy + lane * 4) = y;
}

// First layer of an MLPLayer with a tiny input width (utils/layers.py:6-19 applied to 12-d agent states,
// 5-d goals, 3-d road points, 8-d road types): y = ReLU(LN(W x + b)), W [256,KIN]. One wave per row.
template <int KIN>
__global__ __launch_bounds__(256) void in_mlp_kernel(const float* __restrict__ X, int ldx,
                                                     const float* __restrict__ W, const float* __restrict__ bias,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ Y, int ldy, int rows) {
  const int lane = threadIdx.x & 63;
  float w[4][KIN];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < KIN; ++k) w[c][k] = W[(lane * 4 + c) * KIN + k];
  const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + lane * 4);
  const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + lane * 4);
  const f32x4 b = *reinterpret_cast<const f32x4*>(beta + lane * 4);
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    float x[KIN];
#pragma unroll
    for (int k = 0; k < KIN; ++k) x[k] = X[(size_t)row * ldx + k];
    f32x4 v = bb;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < KIN; ++k) v[c] = fmaf(w[c][k], x[k], v[c]);
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
    const f32x4 d = v - mean;
    const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / 256.f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    f32x4 y = d * rstd * g + b;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = fmaxf(y[i], 0.f);
    *reinterpret_cast<f32x4*>(Y + (size_t)row * ldy + lane * 4) = y;
  }
}

// dst[i, :] = src[index[i], :] (gather) or dst[index[i], :] = src[i, :] (scatter); rows of `width` floats
// (width % 4 == 0).  Used to pull the current-timestep token rows out of / back into the full sequence.
template <bool SCATTER>
__global__ void row_copy_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd,
                                const int* __restrict__ index, int rows, int width4) {
  const int i = blockIdx.x * blockDim.y + threadIdx.y;
  if (i >= rows) return;
  const int r = index[i];
  if (r < 0) return;
  const float* s = SCATTER ? src + (size_t)i * lds_ : src + (size_t)r * lds_;
  float* d = SCATTER ? dst + (size_t)r * ldd : dst + (size_t)i * ldd;
  for (int c = threadIdx.x; c < width4; c += blockDim.x)
    reinterpret_cast<f32x4*>(d)[c] = reinterpret_cast<const f32x4*>(s)[c];
}

#ifndef GEMM_TBK
#define GEMM_TBK 16
#endif
// ------------------------------------------------------------------------------------------------ host launchers
int launch_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr,
                   float* C, int ldc, int M, int N, int K, int relu, hipStream_t st) {
  if (M <= 0) return CTRLSIM_OK;
  if (K % 32 != 0 || (lda & 3) || (ldw & 3) || N <= 0) return CTRLSIM_EINVAL;
  const int m_tiles = (M + BM - 1) / BM, n_tiles = (N + BN - 1) / BN;
  const int total = ((m_tiles + 7) / 8) * 8 * n_tiles;
  const int resident = 256 * (GEMM_TBK == 16 ? 3 : 2);          // CUs x workgroups per CU (LDS / VGPR bound)
  const int grid = total < resident ? total : resident;
  dim3 g(grid), b(256);
  prof_before(PROF_GEMM, st);
  if (R) {
    if (relu) return CTRLSIM_EINVAL;
    hipLaunchKernelGGL((gemm_nt_f32_kernel<GEMM_TBK, false, true>), g, b, 0, st, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K,
                       m_tiles, n_tiles);
  } else if (relu) {
    hipLaunchKernelGGL((gemm_nt_f32_kernel<GEMM_TBK, true, false>), g, b, 0, st, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K,
                       m_tiles, n_tiles);
  } else {
    hipLaunchKernelGGL((gemm_nt_f32_kernel<GEMM_TBK, false, false>), g, b, 0, st, A, lda, W, ldw, bias, R, ldr, C, ldc, M, N,
                       K, m_tiles, n_tiles);
  }
  prof_after(PROF_GEMM, 2.0 * (double)M * (double)N * (double)K, st,
             4.0 * ((double)M * K + (double)M * N * (R ? 2.0 : 1.0) + (double)N * K));
  return ctrlsim_launch_status();
}

int launch_layernorm256(const float* X, int ldx, const float* Radd, int ldr, const float* gamma, const float* beta,
                        float* Y, int ldy, int rows, int relu, hipStream_t st) {
  if (rows <= 0) return CTRLSIM_OK;
  dim3 g((rows + 3) / 4), b(256);
  if (relu)
    hipLaunchKernelGGL((layernorm256_kernel<true>), g, b, 0, st, X, ldx, Radd, ldr, gamma, beta, Y, ldy, rows, ctrlsim_nonfinite_ptr());
  else
    hipLaunchKernelGGL((layernorm256_kernel<false>), g, b, 0, st, X, ldx, Radd, ldr, gamma, beta, Y, ldy, rows, ctrlsim_nonfinite_ptr());
  return ctrlsim_launch_status();
}

int launch_in_mlp(const float* X, int ldx, int kin, const float* W, const float* bias, const float* gamma,
                  const float* beta, float* Y, int ldy, int rows, hipStream_t st) {
  if (rows <= 0) return CTRLSIM_OK;
  int blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  dim3 g(blocks), b(256);
  switch (kin) {
    case 3: hipLaunchKernelGGL((in_mlp_kernel<3>), g, b, 0, st, X, ldx, W, bias, gamma, beta, Y, ldy, rows); break;
    case 5: hipLaunchKernelGGL((in_mlp_kernel<5>), g, b, 0, st, X, ldx, W, bias, gamma, beta, Y, ldy, rows); break;
    case 8: hipLaunchKernelGGL((in_mlp_kernel<8>), g, b, 0, st, X, ldx, W, bias, gamma, beta, Y, ldy, rows); break;
    case 12: hipLaunchKernelGGL((in_mlp_kernel<12>), g, b, 0, st, X, ldx, W, bias, gamma, beta, Y, ldy, rows); break;
    default: return CTRLSIM_EINVAL;
  }
  return ctrlsim_launch_status();
}

int launch_row_copy(const float* src, int lds_, float* dst, int ldd, const int* index, int rows, int width,
                    int scatter, hipStream_t st) {
  if (rows <= 0) return CTRLSIM_OK;
  if (width & 3) return CTRLSIM_EINVAL;
  dim3 b(64, 4), g((rows + 3) / 4);
  if (scatter)
    hipLaunchKernelGGL((row_copy_kernel<true>), g, b, 0, st, src, lds_, dst, ldd, index, rows, width / 4);
  else
    hipLaunchKernelGGL((row_copy_kernel<false>), g, b, 0, st, src, lds_, dst, ldd, index, rows, width / 4);
  return ctrlsim_launch_status();
}
