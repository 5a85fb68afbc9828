// Known-byte launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950, one kernel per ACCESS PATTERN the hot kernels of this
// repo use (profiles/r05_pmc_traffic.md).  MI355X_MICROARCH.md says FETCH_SIZE reports half the bytes of a wide coalesced streaming read
// and leaves the other widths and WRITE_SIZE uncalibrated; rounds 1-4 used x1 for one kernel family and x2 for another.  Every kernel
// below touches each byte of a buffer that is larger than the Infinity Cache (default 1 GiB) exactly once, so bytes = buffer size:
//   rd16_coalesced   global_load_dwordx4, a wave reads 1 KB contiguous                      (epilogue residual rows, Q rows)
//   rd16_rows        global_load_dwordx4 x2 per lane, lane l: row l & 31, 32 B at column 16 ks + 8 (l >> 5)  (row-stationary A operand: ffn / inproj)
//   rd_ldsdma        global_load_lds_dwordx4, 1 KB per wave instruction                     (weight / K-V image / ws256 row stream)
//   rd4_coalesced    global_load_dword                                                        (index / table reads)
//   wr16_coalesced   global_store_dwordx4                                                     (row-major outputs)
//   wr16_nt          global_store_dwordx4 nt                                                  (ws256 / inproj outputs)
//   wr8_runs         global_store_dwordx2, 64 lanes = one 512-byte run                        (K image entries of inproj_rs)
//   wr4_coalesced    global_store_dword
// build: hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip ; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rd16_coalesced(const f32x4* __restrict__ src, size_t n16, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc += src[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

// rows of 1 KB; a wave owns 32 rows; k-step ks: lane reads floats [16 ks + 8 half, +8) of its row (two 16-byte loads)
__global__ __launch_bounds__(256) void rd16_rows(const float* __restrict__ src, size_t rows, float* sink) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t r0 = wave * 32; r0 + 32 <= rows; r0 += nwaves * 32) {
    const float* p = src + (r0 + l31) * 256 + half * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      acc += *reinterpret_cast<const f32x4*>(p + ks * 16);
      acc += *reinterpret_cast<const f32x4*>(p + ks * 16 + 4);
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}

__global__ __launch_bounds__(256) void rd_ldsdma(const float* __restrict__ src, size_t n16, float* sink) {
  __shared__ __attribute__((aligned(16))) float buf[8 * 1024];          // 32 KB: 8 pieces of 4 KB per workgroup in flight
  const int wave = threadIdx.x >> 6;
  const size_t per_wg = 8 * 256;                                          // 16-byte pieces per workgroup iteration
  for (size_t base = (size_t)blockIdx.x * per_wg; base + per_wg <= n16; base += (size_t)gridDim.x * per_wg) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (base + j * 256 + threadIdx.x) * 4),
                                       (__attribute__((address_space(3))) void*)(buf + j * 1024 + wave * 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);
  }
  __syncthreads();
  if (buf[threadIdx.x] == 12345.678f) *sink = buf[threadIdx.x];
}

__global__ __launch_bounds__(256) void rd4_coalesced(const float* __restrict__ src, size_t n4, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += src[i];
  if (acc == 12345.678f) *sink = acc;
}

__global__ __launch_bounds__(256) void wr16_coalesced(f32x4* __restrict__ dst, size_t n16) {
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = v;
}

__global__ __launch_bounds__(256) void wr16_nt(f32x4* __restrict__ dst, size_t n16) {
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, dst + i);
}

__global__ __launch_bounds__(256) void wr8_runs(f32x2* __restrict__ dst, size_t n8) {
  const f32x2 v = {1.f, (float)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, dst + i);
}

__global__ __launch_bounds__(256) void wr4_coalesced(float* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = (float)threadIdx.x;
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 1024) << 20;     // MiB
  const int reps = argc > 2 ? atoi(argv[2]) : 3;
  float *buf, *sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, bytes));
  const int grid = 256 * 8;
  printf("{\"bytes_per_launch\": %zu, \"launches_per_kernel\": %d}\n", bytes, reps);
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(rd16_coalesced, dim3(grid), dim3(256), 0, 0, (const f32x4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(rd16_rows, dim3(grid), dim3(256), 0, 0, buf, bytes / 1024, sink);
    hipLaunchKernelGGL(rd_ldsdma, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, sink);
    hipLaunchKernelGGL(rd4_coalesced, dim3(grid), dim3(256), 0, 0, buf, bytes / 4, sink);
    hipLaunchKernelGGL(wr16_coalesced, dim3(grid), dim3(256), 0, 0, (f32x4*)buf, bytes / 16);
    hipLaunchKernelGGL(wr16_nt, dim3(grid), dim3(256), 0, 0, (f32x4*)buf, bytes / 16);
    hipLaunchKernelGGL(wr8_runs, dim3(grid), dim3(256), 0, 0, (f32x2*)buf, bytes / 8);
    hipLaunchKernelGGL(wr4_coalesced, dim3(grid), dim3(256), 0, 0, buf, bytes / 4);
    CK(hipDeviceSynchronize());
  }
  CK(hipGetLastError());
  return 0;
}
