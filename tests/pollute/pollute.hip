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

// Victim probe (tools/victim_probe.py): every workgroup fills `lds_bytes` of dynamic LDS and 48 registers per lane with a
// pattern and keeps re-checking both for `ticks` of the 100 MHz counter, with workgroup barriers in between, while other
// kernels' workgroups come and go on its CU.  log: [0] LDS mismatches, [1] register mismatches, then up to 64 records of
// (block, word offset or register index, expected, got, iteration, kind).
extern "C" __global__ __launch_bounds__(256) void victim_kernel(unsigned long long ticks, int lds_words, unsigned seed, unsigned* log, float* gbuf) {
  extern __shared__ unsigned vl[];
  const unsigned tid = threadIdx.x, blk = blockIdx.x;
  auto pat = [&](unsigned i) { return (seed ^ (blk * 0x9E3779B9u)) + i * 0x85EBCA6Bu; };
  for (int i = tid; i < lds_words; i += 256) vl[i] = pat(i);
  unsigned r[48];
#pragma unroll
  for (int k = 0; k < 48; ++k) { r[k] = pat(0x100000u + tid * 64 + k); asm volatile("" : "+v"(r[k])); }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned it = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    for (int i = tid; i < lds_words; i += 256) {
      const unsigned got = vl[i], want = pat(i);
      if (got != want) {
        const unsigned n = atomicAdd(&log[0], 1u);
        if (n < 48) { unsigned* e = log + 8 + 6 * n; e[0] = blk; e[1] = i; e[2] = want; e[3] = got; e[4] = it; e[5] = 0; }
        vl[i] = want;
      }
    }
#pragma unroll
    for (int k = 1; k < 48; ++k) {
      asm volatile("" : "+v"(r[k]));
      const unsigned want = pat(0x100000u + tid * 64 + k);
      if (r[k] != want) {
        const unsigned n = atomicAdd(&log[1], 1u);
        if (n < 16) { unsigned* e = log + 8 + 6 * (48 + n); e[0] = blk; e[1] = tid * 64 + k; e[2] = want; e[3] = r[k]; e[4] = it; e[5] = 1; }
        r[k] = want;
      }
    }
    // arithmetic check: a chain of the operations the simulator's narrow phase uses (sqrt, IEEE division, sin / cos, fma) on
    // fixed inputs must give the same bits every time
    {
      float acc = 0.f;
#pragma unroll 4
      for (int k = 0; k < 16; ++k) {
        const float a = 1.0f + (float)((pat(k + tid) >> 8) & 0xFFFF) * (1.0f / 1024.0f);
        const float c = sinf(a) * cosf(a * 0.37f) + sqrtf(a) / (a + 0.25f);
        acc = fmaf(acc, 0.75f, c);
      }
      const unsigned bits = __float_as_uint(acc);
      if (it == 0) r[0] = bits ^ pat(0x100000u + tid * 64);       // remember (register slot 0 holds pattern ^ first result)
      else if ((r[0] ^ pat(0x100000u + tid * 64)) != bits) {
        const unsigned n = atomicAdd(&log[3], 1u);
        if (n < 16) { unsigned* e = log + 8 + 6 * (48 + n); e[0] = blk; e[1] = tid; e[2] = r[0] ^ pat(0x100000u + tid * 64); e[3] = bits; e[4] = it; e[5] = 2; }
      }
    }
    // global-memory hand-over between the waves of the workgroup, the way sim_step's contact records travel: thread t writes
    // 20-float records t, t + 256, ... of the block's slab, a barrier, then reads the records of thread t + 64 (the next wave)
    if (gbuf) {
      float* slab = gbuf + (size_t)blk * 2048 * 20;
      for (int rec = tid; rec < 2048; rec += 256) {
        float* m = slab + rec * 20;
        float old0 = m[0];
        for (int z = 0; z < 20; ++z) m[z] = (float)(it * 7 + rec + z);
        if (it > 0 && old0 != (float)((it - 1) * 7 + rec)) atomicAdd(&log[5], 1u);
      }
      __syncthreads();
      for (int rec = (tid + 64) & 255; rec < 2048; rec += 256) {
        const float* m = slab + rec * 20;
        bool ok = true;
        for (int z = 0; z < 20; ++z) ok = ok && m[z] == (float)(it * 7 + rec + z);
        if (!ok) {
          const unsigned n = atomicAdd(&log[4], 1u);
          if (n < 8) { unsigned* e = log + 8 + 6 * (56 + n); e[0] = blk; e[1] = rec; e[2] = it; e[3] = __float_as_uint(m[0]); e[4] = __float_as_uint(m[19]); e[5] = 3; }
        }
      }
    }
    __syncthreads();
    ++it;
  }
  if (tid == 0) atomicAdd(&log[2], it);
}
extern "C" int victim_launch(unsigned ms, int blocks, int lds_bytes, unsigned seed, unsigned* log, float* gbuf, hipStream_t st) {
  if (hipFuncSetAttribute((const void*)victim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -1;
  hipLaunchKernelGGL(victim_kernel, dim3(blocks), dim3(256), lds_bytes, st, (unsigned long long)ms * 100000ull, lds_bytes / 4, seed, log, gbuf);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
