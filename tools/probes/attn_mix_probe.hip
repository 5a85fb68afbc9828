// Round 6 probe: what does the INSTRUCTION MIX of the split-operand flash-attention inner loop sustain on one SIMD, with no memory system behind it?
// One 32-key sub-tile of attention_bf16x6_kernel = 6 dependent MFMAs (S^T = K.Q^T, three partial products x two k-steps) -> softmax on 16 scores
// per lane (seed, exp2, row sum, speculative-base test) -> P split into two fp16 planes (v_cvt_pk_f16_f32 + v_fma_mixlo/hi) -> 6 dependent MFMAs
// (O^T += V^T.P^T).  The probe runs exactly that chain on register / LDS resident data, with switches for each ingredient:
//   LDSR   fragment reads from LDS (4 x ds_read_b128 for K, 8 x ds_read_b64 for V per sub-tile) or constant registers
//   VALU   the softmax + split block, or P taken as constant fragments
//   MFMA   the two chains, or none (VALU only)
//   BAR    a workgroup barrier every tile (two sub-tiles), as the staged kernel has
// at 1..6 waves per SIMD (8-wave workgroups, occupancy set by the dynamic LDS size).  Output: cycles per sub-tile and wave (s_memtime),
// matrix-pipe utilisation = 12 MFMA x 32 cycles x waves per SIMD / cycles per sub-tile.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o attn_mix_probe attn_mix_probe.hip && ./attn_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 opx8 __attribute__((ext_vector_type(8)));
typedef _Float16 opx4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)

__device__ __forceinline__ void split_pair(float a, float b, unsigned (&pl)[2]) {
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pl[0]) : "v"(a), "v"(b));
  unsigned lo;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(pl[0]), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(pl[0]), "v"(b));
  pl[1] = lo;
}
__device__ __forceinline__ void split_frag(const float* x, opx8 (&f)[2]) {
  u32x4 w[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned t[2];
    split_pair(x[2 * i], x[2 * i + 1], t);
    w[0][i] = t[0]; w[1][i] = t[1];
  }
  f[0] = __builtin_bit_cast(opx8, w[0]);
  f[1] = __builtin_bit_cast(opx8, w[1]);
}
__device__ __forceinline__ opx8 cat8(opx4 a, opx4 b) { return opx8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

template <bool LDSR, bool VALU, bool MFMAS, bool BAR>
__global__ __launch_bounds__(512, 6) void probe(float* out, unsigned long long* ticks, int iters) {
  extern __shared__ __attribute__((aligned(16))) _Float16 arena[];     // 16 KB tile image, [K planes | V planes]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  for (int i = tid; i < 8192; i += 512) arena[i] = (_Float16)(0.001f * (float)((i * 7 + 3) & 255) - 0.12f);
  __syncthreads();
  opx8 qf[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[a][p][e] = (_Float16)(0.01f * (float)((lane + e + a + 3 * p) & 15) - 0.07f);
  f32x16 oa;
#pragma unroll
  for (int r = 0; r < 16; ++r) oa[r] = 0.f;
  float l_run = 0.f, m_base = 0.25f;
  opx8 kc[2][2], vc[2][2];                                              // constant fragments (LDSR off)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int p = 0; p < 2; ++p) { kc[a][p] = qf[a][p]; vc[a][p] = qf[p][a]; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      opx8 k0f[2], k1f[2];
      if (LDSR) {
        const _Float16* kr_ = arena + (half * 64 + sub * 32 + l31) * 8;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          k0f[p] = *reinterpret_cast<const opx8*>(kr_ + p * 2048);
          k1f[p] = *reinterpret_cast<const opx8*>(kr_ + p * 2048 + 1024);
        }
      } else {
#pragma unroll
        for (int p = 0; p < 2; ++p) { k0f[p] = kc[0][p]; k1f[p] = kc[1][p]; }
      }
      f32x16 s0;
      const float seed = MFMAS ? -m_base : -m_base + l_run * 1e-30f;    // (VALU-only runs: keep the block loop-variant)
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[r] = seed;
      if (MFMAS) {
        s0 = MFMA(k0f[1], qf[0][0], s0); s0 = MFMA(k1f[1], qf[1][0], s0);
        s0 = MFMA(k0f[0], qf[0][1], s0); s0 = MFMA(k1f[0], qf[1][1], s0);
        s0 = MFMA(k0f[0], qf[0][0], s0); s0 = MFMA(k1f[0], qf[1][0], s0);
      }
      opx8 pf[2][2];
      if (VALU) {
        float sc[16], psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(s0[r]); psum += sc[r]; }
        if (__any(!(psum < 32768.f))) {                                  // never taken on this data: the speculative-base test of the kernel
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[r] *= 0.5f;
          m_base += 1.f;
        }
        l_run += psum;
        split_frag(sc, pf[0]);
        split_frag(sc + 8, pf[1]);
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int p = 0; p < 2; ++p) pf[a][p] = qf[a][p];
        if (MFMAS) l_run += s0[0];
      }
      opx8 v0f[2], v1f[2];
      if (LDSR) {
        const _Float16* vr_ = arena + 4096 + ((sub * 8 + half) * 32 + l31) * 4;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const opx4 a0 = *reinterpret_cast<const opx4*>(vr_ + p * 2048);
          const opx4 a1 = *reinterpret_cast<const opx4*>(vr_ + p * 2048 + 256);
          const opx4 b0 = *reinterpret_cast<const opx4*>(vr_ + p * 2048 + 512);
          const opx4 b1 = *reinterpret_cast<const opx4*>(vr_ + p * 2048 + 768);
          v0f[p] = cat8(a0, a1);
          v1f[p] = cat8(b0, b1);
        }
      } else {
#pragma unroll
        for (int p = 0; p < 2; ++p) { v0f[p] = vc[0][p]; v1f[p] = vc[1][p]; }
      }
      if (MFMAS) {
        oa = MFMA(v0f[1], pf[0][0], oa); oa = MFMA(v1f[1], pf[1][0], oa);
        oa = MFMA(v0f[0], pf[0][1], oa); oa = MFMA(v1f[0], pf[1][1], oa);
        oa = MFMA(v0f[0], pf[0][0], oa); oa = MFMA(v1f[0], pf[1][0], oa);
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int p = 0; p < 2; ++p) oa[2 * a + p] += (float)pf[a][p][0] + (float)v0f[p][a];
      }
    }
    if (BAR) __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = l_run;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += oa[r];
  if (s == 123.456f) out[0] = s;
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <bool LDSR, bool VALU, bool MFMAS, bool BAR>
void run(const char* name, float* d_out, unsigned long long* d_ticks, int ncu) {
  const int iters = 2000;
  printf("%-44s", name);
  for (int wg_per_cu = 1; wg_per_cu <= 3; ++wg_per_cu) {          // 8-wave workgroups: 2, 4, 6 waves per SIMD
    const int lds = 160 * 1024 / wg_per_cu - 1024;                  // occupancy through the LDS size
    auto kern = probe<LDSR, VALU, MFMAS, BAR>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = ncu * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, d_out, d_ticks, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> t(grid);
    hipMemcpy(t.data(), d_ticks, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : t) mean += (double)v;
    mean /= grid;
    const double per_sub = mean / (2.0 * iters);                    // ticks per sub-tile of one wave (all waves run the same loop)
    const double util = MFMAS ? 12.0 * 32.0 * (2 * wg_per_cu) / per_sub : 0.0;
    printf("  | %d w/SIMD: %7.0f tick/sub  pipe %4.2f  %6.3f ms", 2 * wg_per_cu, per_sub, util, ms);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  float* d_out; unsigned long long* d_ticks;
  hipMalloc(&d_out, 4096);
  hipMalloc(&d_ticks, 8 * 4096);
  printf("%s, %d CUs; 8-wave workgroups; tick = s_memtime\n", p.name, ncu);
  run<true, true, true, true>("full: LDS reads + softmax/split + MFMA + barrier", d_out, d_ticks, ncu);
  run<true, true, true, false>("no barrier", d_out, d_ticks, ncu);
  run<false, true, true, true>("no LDS reads (+ barrier)", d_out, d_ticks, ncu);
  run<false, true, true, false>("no LDS reads, no barrier", d_out, d_ticks, ncu);
  run<true, false, true, false>("MFMA + LDS reads only (no softmax/split)", d_out, d_ticks, ncu);
  run<false, false, true, false>("MFMA only", d_out, d_ticks, ncu);
  run<true, true, false, false>("softmax/split + LDS reads only (no MFMA)", d_out, d_ticks, ncu);
  run<false, true, false, false>("softmax/split only", d_out, d_ticks, ncu);
  return 0;
}
