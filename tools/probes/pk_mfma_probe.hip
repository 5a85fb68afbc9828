// Probe for the co-residency hazard of DESIGN.md section 4 (second form): do packed-fp32 VALU instructions (v_pk_mul_f32 /
// v_pk_add_f32 / v_pk_fma_f32 — what clang's SLP vectoriser makes of neighbouring scalar float operations) give wrong results when the
// wave shares its CU with waves of ANOTHER kernel that keeps the matrix pipe and the LDS busy?
//   victim:    per iteration the same update is computed twice, with packed instructions and with scalar ones (inline asm, the
//              compiler cannot merge them); any bitwise difference is counted per 16-lane group.  1 workgroup per CU, few VGPRs.
//   aggressor: v_mfma_f32_32x32x16_f16 chains + LDS-DMA + ds_read_b128, 168 VGPRs, 3 workgroups per CU, on a second stream.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_mfma_probe pk_mfma_probe.hip ; run: ./pk_mfma_probe [ms=400]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void victim(unsigned iters, unsigned long long* bad, unsigned long long* hist) {
  __shared__ float pad[13 * 1024];                     // 52 KB like the simulator step: the aggressor fits beside it
  const unsigned tid = threadIdx.x, lane = tid & 63;
  pad[tid] = (float)tid;
  __syncthreads();
  f32x2 p = {1.0f + 0.001f * tid, 0.5f - 0.002f * tid}, q = p;           // packed path / scalar path
  const f32x2 a = {1.0000001f, 0.9999999f}, b = {1e-3f * pad[tid & 255], -2e-3f};
  unsigned long long nbad = 0;
  for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f32x2 t;
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(p), "v"(a));
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(t), "v"(b));
      float s0, s1;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(q[0]), "v"(a[0]));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(q[1]), "v"(a[1]));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(q[0]) : "v"(s0), "v"(b[0]));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(q[1]) : "v"(s1), "v"(b[1]));
    }
    const bool m = __builtin_bit_cast(unsigned, p[0]) != __builtin_bit_cast(unsigned, q[0]) ||
                   __builtin_bit_cast(unsigned, p[1]) != __builtin_bit_cast(unsigned, q[1]);
    if (m) { ++nbad; atomicAdd(&hist[lane >> 4], 1ull); p = q; }
    if ((it & 1023u) == 0) { p[0] = q[0] = 1.0f + 0.001f * tid; p[1] = q[1] = 0.5f - 0.002f * tid; }
  }
  if (nbad) atomicAdd(bad, nbad);
}

// Round 4: the same for v_pk_mov_b32 — the one packed instruction -fno-slp-vectorize removes ENTIRELY from the library (the packed
// arithmetic stays, in plain and in swizzled forms).  Three forms as clang emits them in sim_step: the in-place swap of a register pair
// (dst = src0 = src1, op_sel:[1,0]), a gather from two pairs with op_sel:[1,0], and one with op_sel:[0,1]; each against v_mov_b32 / v_swap_b32.
__global__ __launch_bounds__(256) void victim_mov(unsigned iters, unsigned long long* bad, unsigned long long* hist) {
  __shared__ float pad[13 * 1024];
  const unsigned tid = threadIdx.x, lane = tid & 63;
  pad[tid] = (float)tid;
  __syncthreads();
  f32x2 p = {1.0f + 0.001f * tid, 0.5f - 0.002f * tid}, q = p;
  const f32x2 a = {1.0000001f, 0.9999999f}, b = {1e-3f * pad[tid & 255], -2e-3f};
  unsigned long long nbad = 0;
  for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // packed-move path (the arithmetic itself is scalar in both paths)
      float t0, t1;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(p[0]), "v"(a[0]));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(p[1]), "v"(a[1]));
      f32x2 t = {t0, t1}, g, h;
      asm volatile("v_pk_mov_b32 %0, %0, %0 op_sel:[1,0]" : "+v"(t));                           // swap in place
      asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(g) : "v"(t), "v"(b));          // g = {t.hi, b.lo}
      asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(h) : "v"(t), "v"(b));          // h = {t.lo, b.hi}
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[0]) : "v"(g[0]), "v"(h[1]));                 // t0 + b.hi   (t swapped: t.hi = t0)
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[1]) : "v"(h[0]), "v"(g[1]));                 // t1 + b.lo
      // scalar path
      float s0, s1, u0, u1, w0, w1;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(q[0]), "v"(a[0]));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(q[1]), "v"(a[1]));
      asm volatile("v_swap_b32 %0, %1" : "+v"(s0), "+v"(s1));
      asm volatile("v_mov_b32 %0, %1" : "=v"(u0) : "v"(s1));
      asm volatile("v_mov_b32 %0, %1" : "=v"(u1) : "v"(b[0]));
      asm volatile("v_mov_b32 %0, %1" : "=v"(w0) : "v"(s0));
      asm volatile("v_mov_b32 %0, %1" : "=v"(w1) : "v"(b[1]));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(q[0]) : "v"(u0), "v"(w1));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(q[1]) : "v"(w0), "v"(u1));
    }
    const bool m = __builtin_bit_cast(unsigned, p[0]) != __builtin_bit_cast(unsigned, q[0]) ||
                   __builtin_bit_cast(unsigned, p[1]) != __builtin_bit_cast(unsigned, q[1]);
    if (m) { ++nbad; atomicAdd(&hist[lane >> 4], 1ull); p = q; }
    if ((it & 1023u) == 0) { p[0] = q[0] = 1.0f + 0.001f * tid; p[1] = q[1] = 0.5f - 0.002f * tid; }
  }
  if (nbad) atomicAdd(bad, nbad);
}

// Round 4, after the assembly bisect named the form (profiles/r04_hazard.md): packed arithmetic whose LOW lane reads the HIGH half of a source
// pair — the op_sel forms clang's SLP vectoriser emits in sim_step (op_sel:[1,0]; op_sel:[0,1] op_sel_hi:[1,0] with neg modifiers).
__global__ __launch_bounds__(256) void victim_cross(unsigned iters, unsigned long long* bad, unsigned long long* hist) {
  __shared__ float pad[13 * 1024];
  const unsigned tid = threadIdx.x, lane = tid & 63;
  pad[tid] = (float)tid;
  __syncthreads();
  f32x2 p = {1.0f + 0.001f * tid, 0.5f - 0.002f * tid}, q = p;
  const f32x2 a = {1.0000001f, 0.9999999f}, b = {1e-3f * pad[tid & 255], -2e-3f};
  unsigned long long nbad = 0;
  for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f32x2 t;
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(p), "v"(a));                                       // t = {p.hi a.lo, p.hi a.hi}
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p) : "v"(t), "v"(b));  // p = {t.lo - b.hi, t.hi - b.lo}
      float s0, s1, u0, u1;
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(q[1]), "v"(a[0]));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(q[1]), "v"(a[1]));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u0) : "v"(s0), "v"(b[1]));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u1) : "v"(s1), "v"(b[0]));
      q[0] = u0; q[1] = u1;
    }
    const bool m = __builtin_bit_cast(unsigned, p[0]) != __builtin_bit_cast(unsigned, q[0]) ||
                   __builtin_bit_cast(unsigned, p[1]) != __builtin_bit_cast(unsigned, q[1]);
    if (m) { ++nbad; atomicAdd(&hist[lane >> 4], 1ull); p = q; }
    if ((it & 1023u) == 0) { p[0] = q[0] = 1.0f + 0.001f * tid; p[1] = q[1] = 0.5f - 0.002f * tid; }
  }
  if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(256, 3) void aggressor(const float* __restrict__ src, size_t n_floats, unsigned iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];       // 40 KB
  const unsigned tid = threadIdx.x, wave = tid >> 6;
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  h8 fa = {}, fb = {};
  fa[0] = (_Float16)(1.0f + tid); fb[1] = (_Float16)0.5f;
  for (unsigned it = 0; it < iters; ++it) {
    const size_t base = ((size_t)(blockIdx.x * 977u + it * 131u) * 4096u) % (n_floats - 16384);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* s = src + base + (size_t)(tid + 256 * i) * 4;
      float* dptr = lds + (wave * 64 + 256 * i) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)dptr, 16, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[i], 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    const float4 v = *reinterpret_cast<const float4*>(lds + (tid & 1023) * 4);
    fa[2] = (_Float16)(v.x * 1e-6f);
    __syncthreads();
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][15];
  if (t == 12345.678f) sink[0] = t;
}

int main(int argc, char** argv) {
  const int ms = argc > 1 ? atoi(argv[1]) : 400;
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  unsigned long long *bad, *hist;
  hipMalloc(&bad, 8); hipMalloc(&hist, 32);
  const size_t n = 64u << 20;
  float *src, *sink;
  hipMalloc(&src, n * 4); hipMalloc(&sink, 64);
  hipMemset(src, 1, n * 4);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  for (int arm = 0; arm < 6; ++arm) {
    const int with_aggr = arm & 1, mov = arm >> 1;
    hipMemset(bad, 0, 8); hipMemset(hist, 0, 32);
    hipDeviceSynchronize();
    const unsigned iters = 400000u * (unsigned)ms / 400u;
    if (with_aggr) hipLaunchKernelGGL(aggressor, dim3(cus * 3), dim3(256), 40 * 1024, s2, src, n, iters / (mov ? 16 : 24), sink);
    if (mov == 2) hipLaunchKernelGGL(victim_cross, dim3(cus), dim3(256), 0, s1, iters, bad, hist);
    else if (mov) hipLaunchKernelGGL(victim_mov, dim3(cus), dim3(256), 0, s1, iters, bad, hist);
    else hipLaunchKernelGGL(victim, dim3(cus), dim3(256), 0, s1, iters, bad, hist);
    hipDeviceSynchronize();
    unsigned long long hb = 0, hh[4] = {0, 0, 0, 0};
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hh, hist, 32, hipMemcpyDeviceToHost);
    printf("%s, aggressor %d: %llu mismatching iterations of %u x %d threads (by 16-lane group: %llu %llu %llu %llu)\n", mov == 2 ? "op_sel packed fp32 vs scalar" : mov ? "v_pk_mov_b32 vs v_mov / v_swap" : "packed vs scalar fp32", with_aggr, hb, iters,
           cus * 256, hh[0], hh[1], hh[2], hh[3]);
    fflush(stdout);
  }
  return 0;
}
