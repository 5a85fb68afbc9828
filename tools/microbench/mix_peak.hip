// do MFMA and VALU work of different waves on one SIMD overlap?  alternate NM dependent MFMAs with NV independent VALU FMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NM, int NV>
__global__ __launch_bounds__(256) void kmix(float* out, int iters) {
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f; const float c = 1.0001f, d = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
      asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c), "v"(d));
    }
  }
  float s = v0 + v1 + v2 + v3; for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 123.456f) out[0] = s;
}
template <int NM, int PER>
__global__ __launch_bounds__(256) void kint(float* out, int iters) {
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f; const float c = 1.0001f, d = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < PER / 4; ++j)
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c), "v"(d));
    }
  }
  float s = v0 + v1 + v2 + v3; for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 123.456f) out[0] = s;
}
template <int NM, int PER> void runi(float* d, int wgs, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kint<NM, PER>), dim3(256 * wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-22s waves/SIMD %d: %.3f ms\n", name, wgs, ms);
}
template <int NM, int NV> void run(float* d, int wgs, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kmix<NM, NV>), dim3(256 * wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  double mf = (double)NM * 32 * iters * wgs, va = (double)NV * 4 * iters * wgs;
  printf("%-22s waves/SIMD %d: %.3f ms   (at 2 GHz: mfma pipe %.3f ms, valu@4cyc %.3f ms)\n", name, wgs, ms, mf / 2e6, va / 2e6);
}
int main() {
  float* d; hipMalloc(&d, 4);
  for (int w = 1; w <= 3; w += 2) {
    run<12, 0>(d, w, "12 MFMA only");
    run<0, 96>(d, w, "96 VALU only");
    run<12, 96>(d, w, "12 MFMA + 96 VALU");
    run<0, 192>(d, w, "192 VALU only");
    run<12, 192>(d, w, "12 MFMA + 192 VALU");
    runi<12, 4>(d, w, "12 x (MFMA, 4 VALU)");
    runi<12, 8>(d, w, "12 x (MFMA, 8 VALU)");
    runi<12, 16>(d, w, "12 x (MFMA, 16 VALU)");
  }
  return 0;
}
