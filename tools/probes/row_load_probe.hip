// How fast does a CU load the fused feed-forward kernel's rows?  A wave of that kernel owns 32 rows of 1 KB and reads them "row per lane":
// lane (l31, half) takes 16 bytes of ITS row per instruction — 64 lanes touch 32 different 128-byte lines, 32 bytes of each.  The alternative
// is a coalesced load (64 lanes = 8 rows x 128 contiguous bytes: 8 lines per instruction) followed by a transposition through LDS.
// This probe runs both patterns from an L2-resident region (every workgroup re-reads the same 128 KB, 4 waves per CU like the kernel),
// so that neither HBM bandwidth nor latency hiding by other waves is in the picture: cycles per wave instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o row_load_probe row_load_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ src, float* sink, unsigned long long* cyc, int iters) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const float* base = src + (size_t)(wave * 32) * 256 + (size_t)((it & 3) * 128) * 256;      // 4 x 128 rows = 512 KB region, shared by all workgroups
    f32x4 raw[32];
    if (PAT == 0) {                       // row per lane: k-step ks, lane reads floats [16 ks + 8 half, +8) of row l31
      const float* p = base + (size_t)l31 * 256 + half * 8;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) { raw[2 * ks] = *reinterpret_cast<const f32x4*>(p + ks * 16); raw[2 * ks + 1] = *reinterpret_cast<const f32x4*>(p + ks * 16 + 4); }
    } else {                              // coalesced: instruction i = (chunk c = i >> 2, part j = i & 3): row 8 j + (lane >> 3), floats [32 c + 4 (lane & 7), +4)
      const float* p = base + (size_t)(lane >> 3) * 256 + (lane & 7) * 4;
#pragma unroll
      for (int i = 0; i < 32; ++i) raw[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(8 * (i & 3)) * 256 + 32 * (i >> 2));
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += raw[i];
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
  if (tid == 0) atomicAdd(cyc, t1 - t0);
}

template <int PAT>
void run(const char* name, const float* d, float* sink, unsigned long long* c, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(c, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<PAT>, dim3(256), dim3(256), 140 * 1024, 0, d, sink, c, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    if (rep) printf("%-28s %8.3f ms  %7.1f ns per 32-instruction batch of a wave = %6.1f ns per CU instruction, %6.1f GB/s per CU, %5.2f TB/s chip\n", name, ms,
                    ms * 1e6 / iters, ms * 1e6 / iters / 128, 131072.0 * iters / (ms * 1e6), 131072.0 * iters * 256 / (ms * 1e9));
  }
}

int main() {
  float *d, *sink; unsigned long long* c;
  hipMalloc(&d, 512 * 1024); hipMalloc(&sink, 4); hipMalloc(&c, 8);
  hipMemset(d, 0, 512 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  run<0>("row per lane (32 lines)", d, sink, c, 4000);
  run<1>("coalesced (8 lines)", d, sink, c, 4000);
  return 0;
}
