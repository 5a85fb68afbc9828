// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the CtRL-Sim rollout path.
// Wavefront = 64 lanes.  The shipped matrix kernels evaluate every fp32 product as partial products of 16-bit operand planes on
// v_mfma_f32_32x32x16_f16 / _bf16 with fp32 accumulation (split.h: two fp16 planes / three products, or three bf16 planes / six):
// token parity with the reference's fp32 path rules out plain bf16 / fp8 operands, and the f32-input MFMA (v_mfma_f32_32x32x2_f32,
// gemm.hip / attention.hip) tops out at 157 TFLOP/s — it stays built as the run-time selectable A/B family of the tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CTRLSIM_OK 0
#define CTRLSIM_EINVAL (-22)
#define CTRLSIM_ELAUNCH (-5)

#define HD 32            // head dim (hidden 256 / 8 heads)
#define DM 256           // hidden dim
#define NHEAD 8
#ifndef MAXC
#define MAXC 16          // context size classes per model batch / launch (class tables travel by value in the kernel arguments)
#endif

static inline int ctrlsim_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CTRLSIM_OK : CTRLSIM_ELAUNCH;
}

// Wave-wide (64-lane) reductions on the DPP cross-lane path: four row-local steps (quad_perm xor 1 / xor 2,
// row_half_mirror, row_mirror: every lane then holds its 16-lane row total) and one combine of the four row totals
// through v_readlane.  hipcc lowers __shfl_xor to ds_bpermute_b32 — an LDS round trip per step, 6 dependent ones per
// reduction — which made the LayerNorm epilogue of the GEMM as long as its k-loop.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_f32(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);   // row_half_mirror
  v += dpp_f32<0x140>(v);   // row_mirror
  return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}

// C/D fragment of mfma_f32_32x32x2f32: lane l, register r -> (row, col)
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// splitmix64 finaliser: the counter-based generator shared with ctrlsim_amd/weights.py
__device__ __host__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ---- non-finite guard.  One device counter per process (allocated on first use, ctrlsim_nonfinite_count reads / clears it).
// The samplers count races that no finite logit won; every fused LayerNorm (GEMM / feed-forward epilogues, layernorm256) counts
// rows whose variance is not finite.  The second matters: ReLU (v_max_f32) maps NaN to 0, so a row that overflowed the fp16
// range of the two-plane operand split (csrc/split.h) would otherwise come out of the MLP heads as finite, wrong logits.
// The guard is a PAIR of device words: [0] non-finite events of the model, [1] simulator events (contacts beyond the island solver's
// table).  Readers (engine.py: nonfinite / check_finite; ctrlsim_nonfinite_count) report them as low half (saturated at 65535) and
// high half (saturated at 32767) of one int — the legacy encoding — but the device never adds across the boundary.
int* ctrlsim_nonfinite_ptr();
int* ctrlsim_simguard_ptr();
unsigned long long* ctrlsim_attn_cprof_ptr();     // per-class cycle counters of the causal attention launches (api.hip), null unless enabled
__device__ __forceinline__ void count_nonfinite_row(float var, int lane, int* __restrict__ counter) {
  if (lane == 0 && !(var <= 3.0e38f)) atomicAdd(counter, 1);
}

// ---- optional per-launch HIP-event timing of the two MFMA kernel classes and four satellite kernels (bench.py's roofline numbers).
// Disabled by default (zero overhead); when enabled every GEMM / attention launch is bracketed by two events on the
// launch stream and tagged with its algorithmic FLOPs; nothing synchronises until ctrlsim_prof_collect().
enum { PROF_GEMM = 0, PROF_ATTN = 1,                         // the two MFMA kernel classes
       PROF_CTX = 2, PROF_EMBED = 3, PROF_SIM = 4, PROF_MAP = 5,   // HBM / latency satellites: build_context, assemble_tokens, sim_step, map_pool
       PROF_CLASSES = 6 };
// kernel-level rows inside the two MFMA classes (ctrlsim_prof_collect_sub; bench.py names them): row = 2 * kind + few, few = the
// launch belongs to a few-row section of the forward (last decoder layer on the queried rows, second pass, K/V-cached steps:
// prof_few() brackets them in forward.hip)
enum { PKIND_OTHER = 0, PKIND_GEMM_QKV_KV = 1,  // Linear whose K / V columns leave as split tile images (QKV, memory K/V)
       PKIND_GEMM_LN = 2,                         // Linear + residual + LayerNorm (+ReLU) epilogue, 64 x 256 tiles
       PKIND_GEMM_PLAIN = 3,                      // plain Linear (cross-attention query projection, heads, map / embedding layers)
       PKIND_FFN = 4,                             // fused feed-forward block
       PKIND_ATTN_CAUSAL = 5, PKIND_ATTN_KEYPAD = 6,   // decoder self-attention / scene + cross attention
       PKIND_COUNT = 7, PSUB_COUNT = 2 * PKIND_COUNT };
void prof_few(bool on);
void prof_before(int cls, hipStream_t st);
void prof_after(int cls, double flops, hipStream_t st, double bytes = 0.0, int kind = PKIND_OTHER);   // bytes = compulsory (algorithmic) HBM traffic

// ---- runtime options (ctrlsim_set_option): which MFMA path the matrix kernels take
enum { OPT_ATTN_IMPL = 0, OPT_GEMM_IMPL = 1,   // 0 = f32-input MFMA, 1 = split-bf16 (bf16x6) MFMA
       OPT_GEMM6_TILE = 2,                       // bf16x6 GEMM tile: 0 = auto, 1 = 128x128, 2 = 64x256 (tuning knob)
       OPT_FFN_FUSED = 3,                        // 1 = linear1-ReLU-linear2-residual-LayerNorm as one kernel (ffn_fused.hip); 2 (default) = with the
                                                 // attention out-projection + residual + LayerNorm in front of it as its leading product (+2.8 %); 3 = and the
                                                 // self-attention out-projection + norm1 + cross-attention query projection as one kernel (built and
                                                 // measured in round 5: -0.35 % against 2, profiles/r05_fusion_k256.md; off)
       OPT_SPLIT = 4,                            // operand split of the split-operand kernels (csrc/split.h): 1 = two fp16 planes, three
                                                 // products (default), 0 = three bf16 planes, six products (full fp32 exponent range)
       OPT_RESERVED_5 = 5,                       // (rounds 2-3: map-encoder pooling on the matrix pipe; the kernel lost and was removed in round 4)
       OPT_GEMM_WS = 6,                          // Linear(256 -> 256 G) bit mask (gemm_bf16x6.hip): 1 / 2 = weight-stationary streaming kernel for large / small
                                                 // launches, 4 = also for the K / V-image Linears, 8 = those through the ROW-stationary kernel (round 4), 16 = tall plain Linears too (off)
       OPT_ATTN_TBL = 7,                         // causal self-attention over the token rows: 1 = visibility masks from the per-class table
                                                 // (one v_cndmask per score), 0 = masks built per query in the kernel
       OPT_LAST_KV = 8,                          // last decoder layer of a rollout pass: 1 = in_proj of keys / values only + queries of the queried rows
       OPT_ATTN_DIRECT = 9,                      // few-query attention launches (<= 96 queries per context): 1 = streaming form, one wave per
                                                 // workgroup, K / V fragments straight from the tile images (no LDS staging)
       OPT_COUNT = 10 };
// run-time view of the selected split (dispatch.hip): planes per operand, 16-bit elements per (context, head, tile) K/V image
int split_npl();
inline size_t split_kimg() { return (size_t)2 * split_npl() * 64 * 32; }
int ctrlsim_option(int key);
