// How an fp32 operand is fed to the 16-bit matrix pipe with fp32-class results: it is split into NPL 16-bit planes whose sum
// is the value, and a product is evaluated as the leading partial products with fp32 accumulation inside the MFMA.
//
//   CTRLSIM_F16X3 = 1 (default):  two fp16 planes (2 x 11 significant bits), three products  hi.lo + lo.hi + hi.hi
//       Operand error 2^-23 relative — below the rounding of the fp32 accumulation itself (measured against float64:
//       tests/test_gpu_ops.py) — for HALF the matrix instructions and 2/3 of the plane bytes of the bf16 scheme.  fp16 has a
//       narrow exponent: weights are pre-multiplied by WSCALE = 2^8 at pack time (undone exactly in the epilogues) so that
//       weights down to 5e-4 keep all 22 bits; activations are O(1)-O(1e3) post-LayerNorm quantities (|x| < 65504 is
//       required; residual planes of |x| < 2^-3 become fp16 subnormals: absolute error <= 2^-25).
//   CTRLSIM_F16X3 = 0:  three bf16 planes (3 x 8 bits = the whole fp32 mantissa, fp32's exponent range), six products
//       hi.lo + lo.hi + mid.mid + mid.hi + hi.mid + hi.hi.
//
// Everything that depends on the choice goes through this header: plane count, element / fragment types, the packed convert,
// the product list, the weight scale.  ctrlsim_amd/pack.py asks the library (ctrlsim_split_scheme) and packs accordingly.
#pragma once
#include "common.h"
#include "classes.h"

#ifndef CTRLSIM_F16X3
#define CTRLSIM_F16X3 1
#endif

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Both schemes are built into the library (build.py compiles gemm_bf16x6.hip, ffn_fused.hip and attention_bf16x6.hip once per
// scheme); everything in those files lives in the namespace SPLIT_NS and dispatch.hip picks one at run time (OPT_SPLIT).
#if CTRLSIM_F16X3
#define SPLIT_NS s1
#define NPL 2
#define NPROD 3
#define WSCALE 256.0f
#define WSCALE_INV 0.00390625f
typedef _Float16 op_t;
typedef _Float16 opx8 __attribute__((ext_vector_type(8)));
typedef _Float16 opx4 __attribute__((ext_vector_type(4)));
typedef _Float16 opx2 __attribute__((ext_vector_type(2)));
#define MFMA_OP(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
__device__ __forceinline__ unsigned op_cvt_pk(float a, float b) {       // two values -> one dword of two fp16 (RNE)
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float op_lo(unsigned u) { return (float)__builtin_bit_cast(opx2, u)[0]; }
__device__ __forceinline__ float op_hi(unsigned u) { return (float)__builtin_bit_cast(opx2, u)[1]; }
// partial products, smallest first; A / B are arrays of NPL fragments
#define SPLIT_TERMS(ACC, A, B) \
  ACC = MFMA_OP(A[1], B[0], ACC); \
  ACC = MFMA_OP(A[0], B[1], ACC); \
  ACC = MFMA_OP(A[0], B[0], ACC);
#define PROD_LIST(X) X(1, 0) X(0, 1) X(0, 0)          /* (plane of A, plane of B), smallest product first */
#else
#define SPLIT_NS s0
#define NPL 3
#define NPROD 6
#define WSCALE 1.0f
#define WSCALE_INV 1.0f
typedef __bf16 op_t;
typedef __bf16 opx8 __attribute__((ext_vector_type(8)));
typedef __bf16 opx4 __attribute__((ext_vector_type(4)));
typedef __bf16 opx2 __attribute__((ext_vector_type(2)));
#define MFMA_OP(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
__device__ __forceinline__ unsigned op_cvt_pk(float a, float b) {       // one v_cvt_pk_bf16_f32 converts AND packs two values (RNE)
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float op_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float op_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
#define SPLIT_TERMS(ACC, A, B) \
  ACC = MFMA_OP(A[2], B[0], ACC); \
  ACC = MFMA_OP(A[0], B[2], ACC); \
  ACC = MFMA_OP(A[1], B[1], ACC); \
  ACC = MFMA_OP(A[1], B[0], ACC); \
  ACC = MFMA_OP(A[0], B[1], ACC); \
  ACC = MFMA_OP(A[0], B[0], ACC);
#define PROD_LIST(X) X(2, 0) X(0, 2) X(1, 1) X(1, 0) X(0, 1) X(0, 0)
#endif

// two fp32 values -> NPL dwords, plane p holding the p-th terms of both (first value in the low half)
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&pl)[NPL]) {
#if CTRLSIM_F16X3 && defined(SPLIT_MIX)
  // residual plane straight from the mixed-precision FMA: lo = f16(a - (float)hi), written into the half it belongs to
  // (v_fma_mixlo_f16 / v_fma_mixhi_f16 read hi as fp16 and a as fp32): three instructions per pair instead of six
  pl[0] = op_cvt_pk(a, b);
  unsigned lo;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(pl[0]), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(pl[0]), "v"(b));
  pl[1] = lo;
#else
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    pl[p] = op_cvt_pk(a, b);
    if (p + 1 < NPL) { a -= op_lo(pl[p]); b -= op_hi(pl[p]); }
  }
#endif
}
// four consecutive values -> NPL (two-dword) plane entries
__device__ __forceinline__ void split_quad(const f32x4 x, u32x2 (&pl)[NPL]) {
  unsigned a[NPL], b[NPL];
  split_pair(x[0], x[1], a);
  split_pair(x[2], x[3], b);
#pragma unroll
  for (int p = 0; p < NPL; ++p) pl[p] = u32x2{a[p], b[p]};
}
// eight fp32 values -> NPL eight-element fragments
__device__ __forceinline__ void split_frag(const float* x, opx8 (&f)[NPL]) {
  u32x4 w[NPL];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned t[NPL];
    split_pair(x[2 * i], x[2 * i + 1], t);
#pragma unroll
    for (int p = 0; p < NPL; ++p) w[p][i] = t[p];
  }
#pragma unroll
  for (int p = 0; p < NPL; ++p) f[p] = __builtin_bit_cast(opx8, w[p]);
}
