// Probe for the co-residency hazard of DESIGN.md section 4: does a wide LDS store (ds_write_b64 / ds_write_b128) read its data VGPRs
// late — after a VALU instruction that follows it has already overwritten them — when the CU's LDS data path is kept busy by
// ANOTHER workgroup (LDS-DMA + ds_read_b128 bursts, the pattern of the split-operand GEMM)?
//   victim:    256 threads, per iteration: v = pattern(iter, lane); ds_write_bNN addr, v; IMMEDIATELY overwrite v's registers with
//              garbage (VALU, inline asm the compiler cannot move); s_waitcnt; read back; compare with the pattern.
//   aggressor: persistent workgroups streaming global memory into LDS with global_load_lds_dwordx4 and reading it back with
//              ds_read_b128, on a second stream, sized so that both kernels share every CU.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_war_probe lds_war_probe.hip ; run: ./lds_war_probe [ms=300]
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pat(unsigned it, unsigned lane, unsigned k) { return (it * 2654435761u) ^ (lane * 40503u + k * 0x9E3779B9u) | 1u; }

template <int WIDTH>   // 1: b32, 2: b64, 4: b128
__global__ __launch_bounds__(256) void victim(unsigned iters, unsigned long long* bad, unsigned long long* first_bad_lane_hist) {
  __shared__ __attribute__((aligned(16))) unsigned buf[256 * 4 + 12 * 1024];     // ~52 KB: leaves room for the aggressor beside it
  const unsigned tid = threadIdx.x, lane = tid & 63;
  unsigned long long nbad = 0;
  const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned*)(buf + tid * 4));
  for (unsigned it = 0; it < iters; ++it) {
    unsigned a = pat(it, tid, 0), b = pat(it, tid, 1), c = pat(it, tid, 2), d = pat(it, tid, 3);
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    if (WIDTH == 1) {
      asm volatile("ds_write_b32 %4, %0\n\tv_mov_b32 %0, 0\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b32 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(a), "+v"(b), "=v"(r0), "=v"(r1) : "v"(addr) : "memory");
      nbad += r0 != pat(it, tid, 0);
    } else if (WIDTH == 2) {
      u32x2 v = {a, b}, r;
      asm volatile("ds_write_b64 %2, %0\n\tv_mov_b64 %0, 0\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b64 %1, %2\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(v), "=v"(r) : "v"(addr) : "memory");
      (void)v;
      const bool m = r[0] != pat(it, tid, 0) || r[1] != pat(it, tid, 1);
      nbad += m;
      if (m) atomicAdd(&first_bad_lane_hist[lane >> 4], 1ull);
    } else {
      u32x2 v0 = {a, b}, v1 = {c, d};
      u32x4 r;
      asm volatile("ds_write2_b64 %3, %0, %1 offset1:1\n\tv_mov_b64 %0, 0\n\tv_mov_b64 %1, 0\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b128 %2, %3\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(v0), "+v"(v1), "=v"(r) : "v"(addr) : "memory");
      const bool m = r[0] != pat(it, tid, 0) || r[1] != pat(it, tid, 1) || r[2] != pat(it, tid, 2) || r[3] != pat(it, tid, 3);
      nbad += m;
      if (m) atomicAdd(&first_bad_lane_hist[lane >> 4], 1ull);
    }
    (void)r2; (void)r3;
  }
  if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(256, 3) void aggressor(const float* __restrict__ src, size_t n_floats, unsigned iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];       // 40 KB
  const unsigned tid = threadIdx.x, wave = tid >> 6;
  float acc = 0.f;
  for (unsigned it = 0; it < iters; ++it) {
    const size_t base = ((size_t)(blockIdx.x * 977u + it * 131u) * 4096u) % (n_floats - 16384);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* s = src + base + (size_t)(tid + 256 * i) * 4;
      float* dptr = lds + (wave * 64 + 256 * i) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)dptr, 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);       // vmcnt(0)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(lds + ((tid + 256 * i) & 2047) * 4);
      acc += v.x + v.y + v.z + v.w;
    }
    __syncthreads();
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int ms = argc > 1 ? atoi(argv[1]) : 300;
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  unsigned long long *bad, *hist;
  hipMalloc(&bad, 8); hipMalloc(&hist, 4 * 8);
  const size_t n = 64u << 20;
  float *src, *sink;
  hipMalloc(&src, n * 4); hipMalloc(&sink, 64);
  hipMemset(src, 1, n * 4);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  for (int with_aggr = 0; with_aggr < 2; ++with_aggr)
    for (int kind = 0; kind < 3; ++kind) {
      hipMemset(bad, 0, 8); hipMemset(hist, 0, 32);
      hipDeviceSynchronize();
      const unsigned iters = 200000u * (unsigned)ms / 300u;
      if (with_aggr) hipLaunchKernelGGL(aggressor, dim3(cus * 2), dim3(256), 40 * 1024, s2, src, n, iters / 8, sink);
      switch (kind) {
        case 0: hipLaunchKernelGGL(victim<1>, dim3(cus), dim3(256), 0, s1, iters, bad, hist); break;
        case 1: hipLaunchKernelGGL(victim<2>, dim3(cus), dim3(256), 0, s1, iters, bad, hist); break;
        case 2: hipLaunchKernelGGL(victim<4>, dim3(cus), dim3(256), 0, s1, iters, bad, hist); break;
        default: break;
      }
      hipDeviceSynchronize();
      unsigned long long hb = 0, hh[4] = {0, 0, 0, 0};
      hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hh, hist, 32, hipMemcpyDeviceToHost);
      const char* names[4] = {"ds_write_b32 + overwrite", "ds_write_b64 + overwrite", "ds_write2_b64 + overwrite", ""};
      printf("%-28s aggressor %d: %llu mismatches in %u iterations x %d threads (by 16-lane group: %llu %llu %llu %llu)\n", names[kind], with_aggr,
             hb, iters, cus * 256, hh[0], hh[1], hh[2], hh[3]);
      fflush(stdout);
    }
  return 0;
}
