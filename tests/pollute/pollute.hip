// Test infrastructure (tests/test_gpu_sim_ctx.py, tools/stress_streams.py; NOT part of libctrlsim_hip.so): a kernel that leaves NaN / huge-integer patterns in every VGPR it can get and in
// 64 KB of LDS, then exits.  Launched in a loop on its own stream during a rollout, it makes any kernel that reads a register
// or LDS word it never wrote (fresh waves inherit whatever the previous occupant left) produce run-dependent results.
// Built by tests/gpu_utils.py::pollute_lib (hipcc --offload-arch=gfx950 -O3 -shared -fPIC) into tests/pollute/libpollute.so.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(64) void pollute_kernel(unsigned* sink, unsigned seed) {
  extern __shared__ unsigned lds[];
  unsigned v[384];
#pragma unroll
  for (int i = 0; i < 384; ++i) v[i] = 0x7fc00000u ^ (seed * 2654435761u + i * 40503u + threadIdx.x);
#pragma unroll
  for (int i = 0; i < 384; ++i) asm volatile("" : "+v"(v[i]));
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = 0xffc12345u ^ (seed + i);
  __syncthreads();
  unsigned a = lds[(threadIdx.x * 97 + seed) & 16383];
#pragma unroll
  for (int i = 0; i < 384; ++i) a += v[i];
  if (a == 0x12345u) sink[0] = a;
}
extern "C" int pollute_launch(unsigned* sink, unsigned seed, int blocks, hipStream_t st) {
  static bool ok = hipFuncSetAttribute((const void*)pollute_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) == hipSuccess;
  if (!ok) return -1;
  hipLaunchKernelGGL(pollute_kernel, dim3(blocks), dim3(64), 65536, st, sink, seed);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// one wave that waits `ticks` of the 100 MHz real-time counter: delays whatever follows it on its stream (tools/stress_streams.py)
extern "C" __global__ void spin_kernel(unsigned long long ticks, unsigned* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (ticks == ~0ull) sink[1] = 1;
}
extern "C" int spin_launch(unsigned us, unsigned* sink, hipStream_t st) {
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, (unsigned long long)us * 100ull, sink);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
