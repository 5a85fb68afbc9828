// Run-time choice between the two builds of the split-operand kernels (csrc/split.h): namespace s1 = two fp16 planes / three MFMA
// products per fp32 product (weights pre-scaled by 2^8; operands must stay inside the fp16 range), s0 = three bf16 planes / six
// products (the whole fp32 exponent range, half the speed).  ctrlsim_set_option(OPT_SPLIT, 1 | 0) selects; the engine falls back
// from s1 to s0 when a rollout produced non-finite logits (engine.py).  Weight planes of both schemes travel in the packed buffer.
#include "common.h"
#include "classes.h"

#define SPLIT_LAUNCHERS(NS)                                                                                                       \
  namespace NS {                                                                                                                  \
  int launch_gemm_nt_bf16x6(const float*, int, const void*, int, int, const float*, const float*, int, float*, int, int, int, int, \
                            int, const float*, const float*, hipStream_t);                                                       \
  int launch_gemm_nt_bf16x6_kv(const float*, int, const void*, int, int, const float*, const float*, int, float*, int, int, int,  \
                               int, int, const float*, const float*, void*, int, int, int, int, int, hipStream_t);                \
  int launch_gemm_nt_bf16x6_kvc(const float*, int, const void*, int, int, const float*, const float*, int, float*, int, int, int, \
                                int, int, const float*, const float*, void*, int, int, const KvClassHost*, hipStream_t);          \
  int launch_inproj_rs(const float*, int, const void*, const float*, float*, int, int, int, void*, int, int, const KvClassHost*,  \
                       hipStream_t);                                                                                              \
  int launch_gemm256_rows(const float*, int, const void*, int, int, const float*, float*, int, const int*, int, hipStream_t);     \
  int launch_ffn_fused_bf16x6(const float*, int, const void*, const float*, const void*, const float*, const float*, const float*, \
                              float*, int, int, int, hipStream_t);                                                               \
  int launch_outproj_ln_q(const float*, int, const float*, int, const void*, const float*, const float*, const float*, const void*,       \
                          const float*, float*, int, float*, int, int, hipStream_t);                                                       \
  int launch_ffn_fused_pre(const float*, int, const float*, int, const void*, const float*, const float*, const float*, const void*,      \
                           const float*, const void*, const float*, const float*, const float*, float*, int, int, int, hipStream_t);     \
  int launch_attention_bf16x6(int, const float*, int, long, const float*, const float*, int, long, float*, int, long, const int*, \
                              const unsigned char*, int, int, int, int, hipStream_t);                                            \
  int launch_attention_bf16x6_pre(int, const float*, int, long, const void*, int, float*, int, long, const int*,                  \
                                  const unsigned char*, int, int, int, int, int, int, int, const void*, hipStream_t);            \
  int launch_attn_mask_tables(int, const AttnClassHost*, hipStream_t);                                                            \
  size_t attn_mask_table_bytes(int, int);                                                                                         \
  int launch_attention_classes(int, const float*, int, const void*, float*, int, const unsigned char*, int, const AttnClassHost*, \
                               hipStream_t);                                                                                      \
  int launch_kv_split(const float*, const float*, int, long, int, int, int, void*, hipStream_t);                                  \
  int launch_kv_split_rows(const float*, const float*, int, long, const int*, int, int, int, void*, hipStream_t);                 \
  int launch_kv_split_rows_classes(const float*, const float*, int, int, const KvRowsHost*, void*, hipStream_t);                  \
  int launch_kv_zero_tail(int, int, int, int, void*, hipStream_t);                                                                \
  int launch_kv_zero_tails(int, const KvTailHost*, int, void* const*, hipStream_t);                                               \
  }
SPLIT_LAUNCHERS(s1)
SPLIT_LAUNCHERS(s0)

int split_npl() { return ctrlsim_option(OPT_SPLIT) ? 2 : 3; }
#define PICK(CALL) (ctrlsim_option(OPT_SPLIT) ? s1::CALL : s0::CALL)

int launch_gemm_nt_bf16x6(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, const float* R, int ldr,
                          float* C, int ldc, int M, int N, int K, int relu, const float* g, const float* b, hipStream_t st) {
  return PICK(launch_gemm_nt_bf16x6(A, lda, W3, n_total, n0, bias, R, ldr, C, ldc, M, N, K, relu, g, b, st));
}
int launch_gemm_nt_bf16x6_kv(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, const float* R, int ldr,
                             float* C, int ldc, int M, int N, int K, int relu, const float* g, const float* b, void* img, int L,
                             int nkt, int col0, int Lreg, int rep_k0, hipStream_t st) {
  return PICK(launch_gemm_nt_bf16x6_kv(A, lda, W3, n_total, n0, bias, R, ldr, C, ldc, M, N, K, relu, g, b, img, L, nkt, col0, Lreg,
                                       rep_k0, st));
}
int launch_gemm_nt_bf16x6_kvc(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, const float* R, int ldr,
                              float* C, int ldc, int M, int N, int K, int relu, const float* g, const float* b, void* img, int col0,
                              int n, const KvClassHost* cls, hipStream_t st) {
  return PICK(launch_gemm_nt_bf16x6_kvc(A, lda, W3, n_total, n0, bias, R, ldr, C, ldc, M, N, K, relu, g, b, img, col0, n, cls, st));
}
int launch_inproj_rs(const float* A, int lda, const void* Wblk, const float* bias, float* C, int ldc, int M, int N, void* img, int col0,
                     int n, const KvClassHost* cls, hipStream_t st) {
  return PICK(launch_inproj_rs(A, lda, Wblk, bias, C, ldc, M, N, img, col0, n, cls, st));
}
int launch_gemm256_rows(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, float* C, int ldc,
                        const int* c_rows, int M, hipStream_t st) {
  return PICK(launch_gemm256_rows(A, lda, W3, n_total, n0, bias, C, ldc, c_rows, M, st));
}
int launch_outproj_ln_q(const float* O, int ldo, const float* R, int ldr, const void* Wop, const float* bo, const float* g0, const float* be0,
                        const void* Wqp, const float* bq, float* X1, int ldx1, float* Q, int ldq, int M, hipStream_t st) {
  return PICK(launch_outproj_ln_q(O, ldo, R, ldr, Wop, bo, g0, be0, Wqp, bq, X1, ldx1, Q, ldq, M, st));
}
int launch_ffn_fused_pre(const float* O, int ldo, const float* R, int ldr, const void* Wop, const float* bo, const float* g0, const float* be0,
                         const void* W1q, const float* b1, const void* W2p, const float* b2, const float* g, const float* b, float* Y, int ldy,
                         int M, int F, hipStream_t st) {
  return PICK(launch_ffn_fused_pre(O, ldo, R, ldr, Wop, bo, g0, be0, W1q, b1, W2p, b2, g, b, Y, ldy, M, F, st));
}
int launch_ffn_fused_bf16x6(const float* X, int ldx, const void* W1p, const float* b1, const void* W2p, const float* b2,
                            const float* g, const float* b, float* Y, int ldy, int M, int F, hipStream_t st) {
  return PICK(launch_ffn_fused_bf16x6(X, ldx, W1p, b1, W2p, b2, g, b, Y, ldy, M, F, st));
}
int launch_attention_bf16x6(int mode, const float* Q, int ldq, long qbs, const float* K, const float* V, int ldkv, long kbs, float* O,
                            int ldo, long obs, const int* q_pos, const unsigned char* key_pad, int B, int Lq, int Lk, int A,
                            hipStream_t st) {
  return PICK(launch_attention_bf16x6(mode, Q, ldq, qbs, K, V, ldkv, kbs, O, ldo, obs, q_pos, key_pad, B, Lq, Lk, A, st));
}
int launch_attention_bf16x6_pre(int mode, const float* Q, int ldq, long qbs, const void* img, int nkt, float* O, int ldo, long obs,
                                const int* q_pos, const unsigned char* key_pad, int B, int Lq, int Lk, int A, int rep_keys,
                                int rep_mult, int rep_pos0, const void* mask_tbl, hipStream_t st) {
  return PICK(launch_attention_bf16x6_pre(mode, Q, ldq, qbs, img, nkt, O, ldo, obs, q_pos, key_pad, B, Lq, Lk, A, rep_keys, rep_mult,
                                          rep_pos0, mask_tbl, st));
}
int launch_attn_mask_tables(int n, const AttnClassHost* cls, hipStream_t st) { return PICK(launch_attn_mask_tables(n, cls, st)); }
size_t attn_mask_table_bytes(int Lq, int nkt) { return s1::attn_mask_table_bytes(Lq, nkt); }   // the table does not depend on the split
int launch_attention_classes(int mode, const float* Q, int ldq, const void* img, float* O, int ldo, const unsigned char* key_pad, int n,
                             const AttnClassHost* cls, hipStream_t st) {
  return PICK(launch_attention_classes(mode, Q, ldq, img, O, ldo, key_pad, n, cls, st));
}
int launch_kv_split(const float* K, const float* V, int ldkv, long kbs, int B, int Lk, int nkt, void* img, hipStream_t st) {
  return PICK(launch_kv_split(K, V, ldkv, kbs, B, Lk, nkt, img, st));
}
int launch_kv_split_rows(const float* K, const float* V, int ldkv, long kbs, const int* pos, int B, int R, int nkt, void* img,
                         hipStream_t st) {
  return PICK(launch_kv_split_rows(K, V, ldkv, kbs, pos, B, R, nkt, img, st));
}
int launch_kv_split_rows_classes(const float* K, const float* V, int ldkv, int n, const KvRowsHost* cls, void* img, hipStream_t st) {
  return PICK(launch_kv_split_rows_classes(K, V, ldkv, n, cls, img, st));
}
int launch_kv_zero_tail(int B, int key0, int n, int nkt, void* img, hipStream_t st) {
  return PICK(launch_kv_zero_tail(B, key0, n, nkt, img, st));
}
int launch_kv_zero_tails(int n, const KvTailHost* t, int nimg, void* const* imgs, hipStream_t st) {
  return PICK(launch_kv_zero_tails(n, t, nimg, imgs, st));
}
