// Round 6 probe 2: does the split-operand flash-attention inner loop gain from TWO 32-query groups per wave (QG = 2) that share the K / V^T
// fragments a wave reads from LDS?  attn_mix_probe + job r06_c said: the real kernel loses 16-22 % to its LDS-DMA (a deeper ring does not
// bring it back) and the fragment reads alone cost ~18 % of the loop — a wave re-reads the whole 16 KB tile from LDS for its 32 queries
// (8 KB per 12 MFMAs).  With QG = 2 a wave holds two Q^T fragment sets, two score tiles and two output accumulators: half the LDS bytes and
// half the waves per MFMA, and two independent MFMA -> softmax -> MFMA chains inside one in-order instruction stream.
// Same work per workgroup in every configuration: 256 queries x 64-key tiles, one barrier per tile, LDS-resident tile (no DMA).
//   QG = 1: 8 waves / workgroup (the shipped shape), QG = 2: 4 waves / workgroup;  1, 2, 3 workgroups per CU (occupancy through the LDS size)
// Output: milliseconds per launch (identical MFMA counts) and MFMAs per microsecond and CU.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o attn_qg_probe attn_qg_probe.hip && ./attn_qg_probe
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

// ORDER 0: group by group (QK_a, softmax_a, PV_a, QK_b, ...); 1: both score chains first, then softmax_a, PV_a, softmax_b, PV_b
template <int QG, int ORDER>
__global__ __launch_bounds__(QG == 1 ? 512 : 256, QG == 1 ? 6 : 3) void probe(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) _Float16 arena[];     // 2 x 16 KB tile images, [K planes | V planes]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  for (int i = tid; i < 16384; i += blockDim.x) arena[i] = (_Float16)(0.001f * (float)((i * 7 + 3) & 255) - 0.12f);
  __syncthreads();
  opx8 qf[QG][2][2];
#pragma unroll
  for (int g = 0; g < QG; ++g)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[g][a][p][e] = (_Float16)(0.01f * (float)((lane + e + a + 3 * p + 5 * g) & 15) - 0.07f);
  f32x16 oa[QG];
  float l_run[QG], m_base = 0.25f;
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    l_run[g] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) oa[g][r] = 0.f;
  }
  for (int it = 0; it < iters; ++it) {
    const _Float16* tile = arena + (it & 1) * 8192;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      opx8 k0f[2], k1f[2];
      const _Float16* kr_ = tile + (half * 64 + sub * 32 + l31) * 8;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        k0f[p] = *reinterpret_cast<const opx8*>(kr_ + p * 2048);
        k1f[p] = *reinterpret_cast<const opx8*>(kr_ + p * 2048 + 1024);
      }
      opx8 v0f[2], v1f[2];
      const _Float16* vr_ = tile + 4096 + ((sub * 8 + half) * 32 + l31) * 4;
      auto load_v = [&]() {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const opx4 a0 = *reinterpret_cast<const opx4*>(vr_ + p * 2048);
          const opx4 a1 = *reinterpret_cast<const opx4*>(vr_ + p * 2048 + 256);
          const opx4 b0 = *reinterpret_cast<const opx4*>(vr_ + p * 2048 + 512);
          const opx4 b1 = *reinterpret_cast<const opx4*>(vr_ + p * 2048 + 768);
          v0f[p] = cat8(a0, a1);
          v1f[p] = cat8(b0, b1);
        }
      };
      f32x16 s0[QG];
      auto qk = [&](int g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[g][r] = -m_base;
        s0[g] = MFMA(k0f[1], qf[g][0][0], s0[g]); s0[g] = MFMA(k1f[1], qf[g][1][0], s0[g]);
        s0[g] = MFMA(k0f[0], qf[g][0][1], s0[g]); s0[g] = MFMA(k1f[0], qf[g][1][1], s0[g]);
        s0[g] = MFMA(k0f[0], qf[g][0][0], s0[g]); s0[g] = MFMA(k1f[0], qf[g][1][0], s0[g]);
      };
      auto soft_pv = [&](int g) {
        float sc[16], psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(s0[g][r]); psum += sc[r]; }
        if (__any(!(psum < 32768.f))) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[r] *= 0.5f;
          m_base += 1.f;
        }
        l_run[g] += psum;
        opx8 pf[2][2];
        split_frag(sc, pf[0]);
        split_frag(sc + 8, pf[1]);
        oa[g] = MFMA(v0f[1], pf[0][0], oa[g]); oa[g] = MFMA(v1f[1], pf[1][0], oa[g]);
        oa[g] = MFMA(v0f[0], pf[0][1], oa[g]); oa[g] = MFMA(v1f[0], pf[1][1], oa[g]);
        oa[g] = MFMA(v0f[0], pf[0][0], oa[g]); oa[g] = MFMA(v1f[0], pf[1][0], oa[g]);
      };
      if (ORDER == 0 || QG == 1) {
        load_v();
#pragma unroll
        for (int g = 0; g < QG; ++g) { qk(g); soft_pv(g); }
      } else {
#pragma unroll
        for (int g = 0; g < QG; ++g) qk(g);
        load_v();
#pragma unroll
        for (int g = 0; g < QG; ++g) soft_pv(g);
      }
    }
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    s += l_run[g];
#pragma unroll
    for (int r = 0; r < 16; ++r) s += oa[g][r];
  }
  if (s == 123.456f) out[0] = s;
}

template <int QG, int ORDER>
void run(const char* name, float* d_out, int ncu) {
  const int iters = 2000;
  printf("%-52s", name);
  for (int wg_per_cu = 1; wg_per_cu <= 3; ++wg_per_cu) {
    const int lds = 160 * 1024 / wg_per_cu - 1024;
    auto kern = probe<QG, ORDER>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int grid = ncu * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(QG == 1 ? 512 : 256), lds, 0, d_out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    // per workgroup and tile: 8 query groups x 2 sub-tiles x 12 MFMAs
    const double mfma_per_cu = (double)wg_per_cu * iters * 8 * 2 * 12;
    printf("  | %d WG/CU: %6.3f ms  %6.1f MFMA/us/CU", wg_per_cu, ms, mfma_per_cu / (ms * 1e3));
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  float* d_out;
  hipMalloc(&d_out, 4096);
  printf("%d CUs; 256 queries per workgroup and 64-key tile, LDS-resident tile, one barrier per tile\n", ncu);
  printf("(pipe-bound reference: 4 SIMDs / 32 cycles = 0.125 MFMA per cycle and CU = 175 MFMA/us/CU at 1.4 GHz, 300 at 2.4 GHz)\n");
  run<1, 0>("QG=1: 8 waves x 32 queries (shipped shape)", d_out, ncu);
  run<2, 0>("QG=2: 4 waves x 64 queries, group by group", d_out, ncu);
  run<2, 1>("QG=2: 4 waves x 64 queries, both score chains first", d_out, ncu);
  return 0;
}
