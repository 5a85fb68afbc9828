// What does an instruction between two MFMAs cost, one wave per SIMD (the fused feed-forward kernel's regime)?  MI355X_MICROARCH.md quotes
// "+43 cycles for the first extra issue slot between two MFMAs on the SAME accumulator, ~6 between MFMAs on different accumulators".  The
// fused feed-forward block's first product is ONE 48-long dependent chain (hacc) with two ds_read_b128 and LDS-DMA address code between
// every group of three; this probe measures the patterns that could replace it, cycles per MFMA by s_memtime, 256 workgroups x 4 waves:
//   P0  48-chain, nothing between                         P1  48-chain, 2 ds_read_b128 between groups of 3 (today's phase A without the DMA)
//   P2  as P1, TWO accumulators alternating per group     P3  P1 + 4 VALU per group            P4  P2 + 4 VALU per group
//   P5  as P2 with FOUR accumulators                      P6  P4 with the fillers spread: one filler after EACH MFMA (between dependent ones)
//   P7  8 accumulators, group of 3 per accumulator, 2 ds_read + 4 VALU between groups (phase B order, kk-major)
//   P8  phase B as shipped: 6 dependent per accumulator with fillers after every 3
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain_probe mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int P>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];     // 64 KB of operands
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768; i += 256) lds[i] = (_Float16)(0.001f * (float)((i * 2654435761u >> 20) & 1023) - 0.5f);
  __syncthreads();
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f16x8*>(lds + (lane * 4 + i) * 8);
  float v0 = (float)lane, v1 = 1.0001f, v2 = 0.5f, v3 = 0.25f;
  const _Float16* base = lds + lane * 8;
  f16x8 a0 = *reinterpret_cast<const f16x8*>(base), a1 = *reinterpret_cast<const f16x8*>(base + 512);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {                                    // 16 groups of 3 MFMAs = 48 per iteration
      const int A = P == 0 || P == 1 || P == 3 ? 0 : (P == 2 || P == 4 || P == 6) ? (g & 1) : P == 5 ? (g & 3) : P == 7 ? (g & 7) : (g >> 1);
      f16x8 n0, n1;
      if (P != 0) {
        n0 = *reinterpret_cast<const f16x8*>(base + ((g * 1024 + it * 64) & 16383));
        n1 = *reinterpret_cast<const f16x8*>(base + ((g * 1024 + 512 + it * 64) & 16383) + 16384);
      }
      if (P == 6) {
        acc[A] = MF(a1, b[0], acc[A]); SB();
        asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)); SB();
        acc[A] = MF(a0, b[1], acc[A]); SB();
        asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)); SB();
        acc[A] = MF(a0, b[0], acc[A]); SB();
      } else {
        acc[A] = MF(a1, b[g & 3], acc[A]);
        acc[A] = MF(a0, b[(g + 1) & 3], acc[A]);
        acc[A] = MF(a0, b[g & 3], acc[A]);
        SB();
        if (P == 3 || P == 4 || P == 7 || P == 8)
          asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2"
                       : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        SB();
      }
      if (P != 0) { a0 = n0; a1 = n1; }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = v0 + v3;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
  if (lane == 0) atomicAdd(cyc, t1 - t0);
}

template <int P>
void run(const char* name, float* d, unsigned long long* c, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<P>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(c, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<P>, dim3(256), dim3(256), 150 * 1024, 0, d, c, iters);      // 150 KB of LDS: one workgroup per CU
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    if (rep) printf("%-70s %7.2f ms  %6.1f cycles/MFMA (s_memtime)  %6.0f TFLOP/s 16-bit\n", name, ms, (double)h / (1024.0 * iters * 48),
                    2.0 * 32 * 32 * 16 * 48.0 * iters * 1024 / ms / 1e9);
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  float* d; unsigned long long* c;
  hipMalloc(&d, 4); hipMalloc(&c, 8);
  run<0>("P0 48-chain, no fillers", d, c, iters);
  run<1>("P1 48-chain, 2 ds_read_b128 per group of 3", d, c, iters);
  run<2>("P2 two accumulators alternating, 2 ds_read per group", d, c, iters);
  run<5>("P5 four accumulators alternating, 2 ds_read per group", d, c, iters);
  run<3>("P3 48-chain, 2 ds_read + 4 VALU per group", d, c, iters);
  run<4>("P4 two accumulators, 2 ds_read + 4 VALU per group", d, c, iters);
  run<6>("P6 two accumulators, 2 ds_read per group + 2 VALU after EACH mfma", d, c, iters);
  run<7>("P7 eight accumulators round robin, 2 ds_read + 4 VALU per group", d, c, iters);
  run<8>("P8 6 dependent per accumulator, 2 ds_read + 4 VALU per 3", d, c, iters);
  return 0;
}
