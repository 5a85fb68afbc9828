// register-only MFMA loop: what rate does the bf16 matrix pipe actually sustain on this part?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}
int main() {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k32<4>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = 2.0 * 32 * 32 * 16 * 4 * (double)iters * 4 * 256 * wgs_per_cu;
      if (rep) printf("32x32x16 bf16, %d waves/SIMD: %.2f ms  %.0f TFLOP/s\n", wgs_per_cu, ms, fl / ms / 1e9);
    }
  }
  for (int nacc = 1; nacc <= 2; ++nacc) for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    if (nacc == 1) hipLaunchKernelGGL(k32<1>, dim3(256 * 3), dim3(256), 0, 0, d, iters * 4);
    else hipLaunchKernelGGL(k32<2>, dim3(256 * 3), dim3(256), 0, 0, d, iters * 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 32 * 32 * 16 * 4 * (double)iters * 4 * 256 * 3;
    if (rep) printf("32x32x16 bf16 dependent chains=%d, 3 waves/SIMD: %.2f ms  %.0f TFLOP/s\n", nacc, ms, fl / ms / 1e9);
  }
  for (int nacc = 1; nacc <= 2; ++nacc) for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    if (nacc == 1) hipLaunchKernelGGL(k32<1>, dim3(256), dim3(256), 0, 0, d, iters * 4);
    else hipLaunchKernelGGL(k32<2>, dim3(256), dim3(256), 0, 0, d, iters * 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 32 * 32 * 16 * 4 * (double)iters * 4 * 256;
    if (rep) printf("32x32x16 bf16 dependent chains=%d, 1 wave/SIMD: %.2f ms  %.0f TFLOP/s\n", nacc, ms, fl / ms / 1e9);
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k16, dim3(512), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 16 * 16 * 32 * 8 * (double)iters * 4 * 512;
    if (rep) printf("16x16x32 bf16, 2 waves/SIMD: %.2f ms  %.0f TFLOP/s\n", ms, fl / ms / 1e9);
  }
  // long run: does the rate sag (power management)?
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k32<4>, dim3(512), dim3(256), 0, 0, d, iters * 10);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 32 * 32 * 16 * 4 * (double)iters * 10 * 4 * 512;
    printf("long 32x32x16: %.2f ms  %.0f TFLOP/s\n", ms, fl / ms / 1e9);
  }
  return 0;
}
