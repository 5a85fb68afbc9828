// extern "C" exports of the C ABI declared in include/ctrlsim.h (thin wrappers over the kernel launchers).
#include "common.h"
#include "classes.h"
#include "../../include/ctrlsim.h"

int launch_gemm_nt(const float*, int, const float*, int, const float*, const float*, int, float*, int, int, int, int, int,
                   hipStream_t);
int launch_gemm_nt_bf16x6(const float*, int, const void*, int, int, const float*, const float*, int, float*, int, int, int,
                          int, int, const float*, const float*, hipStream_t);
int launch_gemm_nt_bf16x6_kv(const float*, int, const void*, int, int, const float*, const float*, int, float*, int, int, int,
                             int, int, const float*, const float*, void*, int, int, int, int, int, hipStream_t);
int launch_gemm256_rows(const float*, int, const void*, int, int, const float*, float*, int, const int*, int, hipStream_t);
int launch_inproj_rs(const float*, int, const void*, const float*, float*, int, int, int, void*, int, int, const KvClassHost*,
                     hipStream_t);
int launch_layernorm256(const float*, int, const float*, int, const float*, const float*, float*, int, int, int,
                        hipStream_t);
int launch_attention(int, const float*, int, long, const float*, const float*, int, long, float*, int, long, const int*,
                     const unsigned char*, int, int, int, int, hipStream_t);
int launch_kv_split(const float*, const float*, int, long, int, int, int, void*, hipStream_t);
int launch_kv_split_rows(const float*, const float*, int, long, const int*, int, int, int, void*, hipStream_t);
int launch_attention_bf16x6_pre(int, const float*, int, long, const void*, int, float*, int, long, const int*,
                                const unsigned char*, int, int, int, int, int, int, int, const void*, hipStream_t);
int launch_attn_mask_tables(int, const AttnClassHost*, hipStream_t);
size_t attn_mask_table_bytes(int, int);
int launch_ffn_fused_bf16x6(const float*, int, const void*, const float*, const void*, const float*, const float*, const float*,
                            float*, int, int, int, hipStream_t);
int launch_ffn_fused_pre(const float*, int, const float*, int, const void*, const float*, const float*, const float*, const void*, const float*,
                         const void*, const float*, const float*, const float*, float*, int, int, int, hipStream_t);
int launch_outproj_ln_q(const float*, int, const float*, int, const void*, const float*, const float*, const float*, const void*, const float*,
                        float*, int, float*, int, int, hipStream_t);
int launch_sim_init(int, int, int, const float*, const float*, const float*, const unsigned char*, float*, float*,
                    unsigned char*, int, float*, hipStream_t);
int launch_sim_set_position(int, int, const float*, float*, hipStream_t);
int launch_sim_step(int, int, int, const int*, const double*, const double*, const float*, const float*,
                    const unsigned char*, float*, float*, unsigned char*, double*, int, int, float, int, float*, const float*,
                    hipStream_t);
int launch_group_build(int, int, int, int, int, int, double, const float*, const int*, int, unsigned long long*, int*, int*,
                       unsigned long long*, unsigned long long*, int*, int*, unsigned char*, hipStream_t);
int launch_ctx_index(int, int, int, const int*, const int*, const unsigned long long*, const int*, const int*, int*, int*,
                     int*, int*, int*, int*, int*, hipStream_t);
int launch_groups_changed(int, int, const int*, const int*, const unsigned long long*, const int*, const int*,
                          const unsigned long long*, int*, hipStream_t);
struct CtxOut { float *st12, *exist, *goal5; int *act_tok, *rtg_bin, *tstep, *slot_gid; float *road_pts, *road_types; };
int launch_build_context_classes(int, const int*, const int*, const CtxOut*, int, int, int, int, int, int, int, int, int, int, const int*,
                                 const int*, const int*, const unsigned long long*, const float*, const int*, const int*,
                                 const double*, const float*, const float*, const float*, const int*, hipStream_t);
int launch_build_context(int, int, int, int, int, int, int, int, int, int, int, int, const int*, const int*, const int*,
                         const unsigned long long*, const float*, const int*, const int*, const double*, const float*,
                         const float*, const float*, const int*, CtxOut, hipStream_t);
int launch_sample_rtg(const float*, int, int, const int*, const int*, const int*, const unsigned char*, const double*, const double*,
                      const float*, uint64_t, const int64_t*, int, int*, int, int, int, hipStream_t);
int launch_group_size_hist(int, int, const int*, const unsigned long long*, int, const int*, int*, hipStream_t);
int launch_ctx_index_classes(int, int, int, int, const int*, const unsigned long long*, const int*, const int*, int, const int*, int*,
                             int*, int*, int*, int*, int*, int*, int*, hipStream_t);
int launch_sample_action(const float*, int, int, const int*, const int*, const int*, float, double, const float*, uint64_t,
                         const int64_t*, int, int*, int*, int, int, int, int, hipStream_t);

#include <vector>
namespace {
struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; hipStream_t st; int sub; };
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t g_pending[PROF_CLASSES];
hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
}  // namespace
void prof_before(int cls, hipStream_t st) {
  if (!g_prof_on) return;
  g_pending[cls] = take_event();
  (void)hipEventRecord(g_pending[cls], st);
}
static bool g_prof_few = false;
void prof_few(bool on) { g_prof_few = on; }
void prof_after(int cls, double flops, hipStream_t st, double bytes, int kind) {
  if (!g_prof_on) return;
  hipEvent_t b = take_event();
  (void)hipEventRecord(b, st);
  g_recs.push_back(ProfRec{g_pending[cls], b, cls, flops, bytes, st, 2 * kind + (g_prof_few ? 1 : 0)});
}

static int g_options[OPT_COUNT] = {1, 1, 0, 2, 1, 0, 15, 1, 1, 1};      // process defaults (ctrlsim_set_option)
// the option table of the ENGINE that bound last (ctrlsim_bind_options): entry >= 0 overrides the process default, -1 inherits it
static int g_bound_options[OPT_COUNT];
static bool g_has_bound_options = false;
// Guard counters (common.h): TWO device words — [0] non-finite events of the model, [1] simulator events — either the pair the CALLER
// bound with ctrlsim_bind (an engine's own 8 bytes of device memory) or, for callers that never bind one, a library-owned pair
// allocated on first use on the then-current device.  Two words, not two halves of one: thousands of non-finite LayerNorm rows of an
// fp16 overflow at production batch sizes must not carry into the simulator's count (round-4 review).
static int* g_guard = nullptr;
static int* own_guard_word() {
  static int* p = nullptr;
  if (!p) {
    if (hipMalloc(reinterpret_cast<void**>(&p), 2 * sizeof(int)) != hipSuccess) { p = nullptr; return nullptr; }
    (void)hipMemset(p, 0, 2 * sizeof(int));
  }
  return p;
}
int* ctrlsim_nonfinite_ptr() { return g_guard ? g_guard : own_guard_word(); }
int* ctrlsim_simguard_ptr() { int* p = ctrlsim_nonfinite_ptr(); return p ? p + 1 : nullptr; }
// events counted in the LIBRARY'S OWN words since the last reset (synchronises the device), in the legacy encoding: the non-finite
// count saturated at 65535 in the low half, the simulator count saturated at 32767 in the high half.  A caller that bound its own
// pair (ctrlsim_bind) reads it itself: this function never touches it, whatever is bound at the moment.
static int nonfinite_count(int reset) {
  int* p = own_guard_word();
  int n[2] = {0, 0};
  if (!p || hipMemcpy(n, p, 2 * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return CTRLSIM_ELAUNCH;
  if (reset && (n[0] || n[1]) && hipMemset(p, 0, 2 * sizeof(int)) != hipSuccess) return CTRLSIM_ELAUNCH;
  const unsigned lo = (unsigned)n[0] > 65535u ? 65535u : (unsigned)n[0], hi = (unsigned)n[1] > 32767u ? 32767u : (unsigned)n[1];
  return (int)(hi * 65536u + lo);
}
// Per-class cycle accounting of the causal self-attention launches over the token rows (profiling runs: bench.py's untimed extra slice).
// 32 slot counts x {workgroup cycles, workgroups}; library-owned device words like the guard counter, allocated on first enable.
static unsigned long long* g_cprof = nullptr;
static bool g_cprof_on = false;
unsigned long long* ctrlsim_attn_cprof_ptr() { return g_cprof_on ? g_cprof : nullptr; }
extern "C" int ctrlsim_attn_class_prof(int enable, unsigned long long* host_out) {
  if (enable) {
    if (!g_cprof && hipMalloc(reinterpret_cast<void**>(&g_cprof), 64 * sizeof(unsigned long long)) != hipSuccess) { g_cprof = nullptr; return CTRLSIM_ELAUNCH; }
    if (hipMemset(g_cprof, 0, 64 * sizeof(unsigned long long)) != hipSuccess) return CTRLSIM_ELAUNCH;
    g_cprof_on = true;
    return CTRLSIM_OK;
  }
  g_cprof_on = false;
  if (host_out && g_cprof && hipMemcpy(host_out, g_cprof, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return CTRLSIM_ELAUNCH;
  return CTRLSIM_OK;
}
extern "C" int ctrlsim_split_scheme() { return ctrlsim_option(OPT_SPLIT) ? 1 : 0; }
extern "C" int ctrlsim_nonfinite_count(int reset) { return nonfinite_count(reset); }
int ctrlsim_option(int key) {
  if (key < 0 || key >= OPT_COUNT) return 0;
  return (g_has_bound_options && g_bound_options[key] >= 0) ? g_bound_options[key] : g_options[key];
}

extern "C" {

// Per-engine state of the library, re-asserted by the caller at the top of every run: the operand split its weight planes / K/V
// images / workspace were built for (0 / 1; -1 = leave) and its guard counter (device int32 it owns and reads itself; NULL = the
// library's own word).  Everything launched until the next bind uses them.
int ctrlsim_bind(int split_scheme, int* guard_counter) {
  if (split_scheme == 0 || split_scheme == 1) g_options[OPT_SPLIT] = split_scheme;
  else if (split_scheme != -1) return CTRLSIM_EINVAL;
  g_guard = guard_counter;
  return CTRLSIM_OK;
}

// the owner of `guard_counter` is about to free it: back to the library's own pair if it is the bound one
int ctrlsim_unbind(const int* guard_counter) {
  if (g_guard == guard_counter) { g_guard = nullptr; g_has_bound_options = false; }   // its option table goes with it
  return CTRLSIM_OK;
}
// the pair bound at the moment (NULL = the library's own): a caller that binds its own for one call restores this afterwards
int* ctrlsim_bound_guard(void) { return g_guard; }

// Per-engine option table: `values` = ctrlsim_option_count() ints (host memory, copied); entry >= 0 overrides the process default of
// ctrlsim_set_option for every launch until the next bind, -1 inherits it.  NULL = back to the process defaults.  Like ctrlsim_bind an
// engine re-asserts its table at the top of every run, so engines with different kernel options take turns in one process.
int ctrlsim_option_count(void) { return OPT_COUNT; }
int ctrlsim_bind_options(const int* values) {
  g_has_bound_options = values != nullptr;
  if (values)
    for (int k = 0; k < OPT_COUNT; ++k) g_bound_options[k] = values[k];
  // the operand split is NOT an option of a table: it is bound with the weight planes / K/V images / workspace it was built for
  // (ctrlsim_bind); a table entry for it would dispatch the other scheme's kernels on this engine's images
  g_bound_options[OPT_SPLIT] = -1;
  return CTRLSIM_OK;
}
int ctrlsim_get_option(int key) { return (key < 0 || key >= OPT_COUNT) ? CTRLSIM_EINVAL : ctrlsim_option(key); }

int ctrlsim_set_option(int key, int value) {
  if (key < 0 || key >= OPT_COUNT) return CTRLSIM_EINVAL;
  g_options[key] = value;
  return CTRLSIM_OK;
}

int ctrlsim_prof_classes(void) { return PROF_CLASSES; }
// enable/disable event timing; enabling clears previous records
void ctrlsim_prof_enable(int on) {
  for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
  g_recs.clear();
  g_prof_on = on != 0;
}
// after the caller synchronised the stream(s): per class total milliseconds, launch count, algorithmic FLOPs
int ctrlsim_prof_collect(double* ms, int64_t* count, double* flops) {
  for (int c = 0; c < PROF_CLASSES; ++c) { ms[c] = 0; count[c] = 0; flops[c] = 0; }
  for (auto& r : g_recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return CTRLSIM_ELAUNCH;
    ms[r.cls] += t; count[r.cls] += 1; flops[r.cls] += r.flops;
  }
  return CTRLSIM_OK;
}

// kernel-level rows (2 * PKIND_* + few, common.h) of the launches on stream st (on_stream != 0) or elsewhere: arrays of ctrlsim_prof_subclasses()
int ctrlsim_prof_subclasses(void) { return PSUB_COUNT; }
int ctrlsim_prof_collect_sub(hipStream_t st, int on_stream, double* ms, int64_t* count, double* flops, double* bytes) {
  for (int c = 0; c < PSUB_COUNT; ++c) { ms[c] = 0; count[c] = 0; flops[c] = 0; bytes[c] = 0; }
  for (auto& r : g_recs) {
    if ((r.st == st) != (on_stream != 0) || r.cls > PROF_ATTN || r.sub < 0 || r.sub >= PSUB_COUNT) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return CTRLSIM_ELAUNCH;
    ms[r.sub] += t; count[r.sub] += 1; flops[r.sub] += r.flops; bytes[r.sub] += r.bytes;
  }
  return CTRLSIM_OK;
}

// the same restricted to the launches ON stream st (on_stream != 0) or on any OTHER stream (on_stream == 0): with several streams
// in flight the intervals of concurrently running kernels overlap, so a caller that wants per-kernel rates of the kernels that
// own the chip (the main stream) separates them from the few-row kernels it deliberately runs underneath them
int ctrlsim_prof_collect_stream(hipStream_t st, int on_stream, double* ms, int64_t* count, double* flops, double* bytes) {
  for (int c = 0; c < PROF_CLASSES; ++c) { ms[c] = 0; count[c] = 0; flops[c] = 0; bytes[c] = 0; }
  for (auto& r : g_recs) {
    if ((r.st == st) != (on_stream != 0)) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return CTRLSIM_ELAUNCH;
    ms[r.cls] += t; count[r.cls] += 1; flops[r.cls] += r.flops; bytes[r.cls] += r.bytes;
  }
  return CTRLSIM_OK;
}

// algorithmic HBM bytes of the recorded launches per class (same records as ctrlsim_prof_collect)
int ctrlsim_prof_bytes(double* bytes) {
  for (int c = 0; c < PROF_CLASSES; ++c) bytes[c] = 0;
  for (auto& r : g_recs) bytes[r.cls] += r.bytes;
  return CTRLSIM_OK;
}

const char* ctrlsim_version(void) { return "ctrlsim-hip 0.1 (gfx950)"; }

int ctrlsim_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, const float* R, int ldr, float* C,
                    int ldc, int M, int N, int K, int relu, hipStream_t st) {
  return launch_gemm_nt(A, lda, W, ldw, bias, R, ldr, C, ldc, M, N, K, relu, st);
}
int ctrlsim_gemm_nt_bf16x6(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, const float* R,
                           int ldr, float* C, int ldc, int M, int N, int K, int relu, const float* ln_gamma,
                           const float* ln_beta, hipStream_t st) {
  return launch_gemm_nt_bf16x6(A, lda, W3, n_total, n0, bias, R, ldr, C, ldc, M, N, K, relu, ln_gamma, ln_beta, st);
}
int ctrlsim_gemm256_rows(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, float* C, int ldc,
                         const int* c_rows, int M, hipStream_t st) {
  return launch_gemm256_rows(A, lda, W3, n_total, n0, bias, C, ldc, c_rows, M, st);
}
int ctrlsim_gemm_nt_kv(const float* A, int lda, const void* W3, int n_total, int n0, const float* bias, float* C, int ldc,
                       int M, int N, int K, void* kv_img, int kv_L, int kv_nkt, int kv_col0, hipStream_t st) {
  return launch_gemm_nt_bf16x6_kv(A, lda, W3, n_total, n0, bias, nullptr, 0, C, ldc, M, N, K, 0, nullptr, nullptr, kv_img, kv_L,
                                  kv_nkt, kv_col0, 0, 0, st);
}
int ctrlsim_gemm_kv_blocks(const float* A, int lda, const void* Wblk, const float* bias, float* C, int ldc, int M, int N,
                           void* kv_img, int kv_L, int kv_nkt, int kv_col0, hipStream_t st) {
  if (!kv_img) return launch_inproj_rs(A, lda, Wblk, bias, C, ldc, M, N, nullptr, N, 0, nullptr, st);     // plain Linear: fp32 rows only
  if (kv_L <= 0 || M % kv_L) return CTRLSIM_EINVAL;
  const KvClassHost c{M / kv_L, kv_L, kv_L, 0, kv_nkt, 0};
  return launch_inproj_rs(A, lda, Wblk, bias, C, ldc, M, N, kv_img, kv_col0, 1, &c, st);
}
int ctrlsim_ffn_fused(const float* X, int ldx, const void* W1p, const float* b1, const void* W2p, const float* b2,
                      const float* gamma, const float* beta, float* Y, int ldy, int M, int F, hipStream_t st) {
  return launch_ffn_fused_bf16x6(X, ldx, W1p, b1, W2p, b2, gamma, beta, Y, ldy, M, F, st);
}
int ctrlsim_ffn_fused_pre(const float* O, int ldo, const float* R, int ldr, const void* Wop, const float* bo, const float* g0, const float* be0,
                          const void* W1q, const float* b1, const void* W2p, const float* b2, const float* gamma, const float* beta, float* Y,
                          int ldy, int M, int F, hipStream_t st) {
  return launch_ffn_fused_pre(O, ldo, R, ldr, Wop, bo, g0, be0, W1q, b1, W2p, b2, gamma, beta, Y, ldy, M, F, st);
}
int ctrlsim_outproj_ln_q(const float* O, int ldo, const float* R, int ldr, const void* Wop, const float* bo, const float* g0, const float* be0,
                         const void* Wqp, const float* bq, float* X1, int ldx1, float* Q, int ldq, int M, hipStream_t st) {
  return launch_outproj_ln_q(O, ldo, R, ldr, Wop, bo, g0, be0, Wqp, bq, X1, ldx1, Q, ldq, M, st);
}
int ctrlsim_layernorm256(const float* X, int ldx, const float* Radd, int ldr, const float* gamma, const float* beta, float* Y,
                         int ldy, int rows, int relu, hipStream_t st) {
  return launch_layernorm256(X, ldx, Radd, ldr, gamma, beta, Y, ldy, rows, relu, st);
}
int ctrlsim_kv_split(const float* K, const float* V, int ldkv, int64_t kbs, const int* pos, int B, int rows, int nkt, void* img,
                     hipStream_t st) {
  if (pos) return launch_kv_split_rows(K, V, ldkv, (long)kbs, pos, B, rows, nkt, img, st);
  return launch_kv_split(K, V, ldkv, (long)kbs, B, rows, nkt, img, st);
}
int ctrlsim_attention_presplit(int mode, const float* Q, int ldq, int64_t qbs, const void* img, int nkt, float* O, int ldo,
                               int64_t obs, const int* q_pos, const uint8_t* key_pad, int B, int Lq, int Lk, int A,
                               hipStream_t st) {
  return launch_attention_bf16x6_pre(mode, Q, ldq, (long)qbs, img, nkt, O, ldo, (long)obs, q_pos, key_pad, B, Lq, Lk, A, 0, 1, Lk,
                                     nullptr, st);
}
int ctrlsim_attention_compact(const float* Q, int ldq, int64_t qbs, const void* img, int nkt, float* O, int ldo, int64_t obs,
                              const int* q_pos, int B, int Lq, int Lk, int A, int rep_keys, int rep_mult, int rep_pos0,
                              hipStream_t st) {
  return launch_attention_bf16x6_pre(1, Q, ldq, (long)qbs, img, nkt, O, ldo, (long)obs, q_pos, nullptr, B, Lq, Lk, A, rep_keys,
                                     rep_mult, rep_pos0, nullptr, st);
}
int64_t ctrlsim_attention_mask_table_bytes(int Lq, int nkt) {
  return (Lq > 0 && nkt > 0) ? (int64_t)attn_mask_table_bytes(Lq, nkt) : CTRLSIM_EINVAL;
}
int ctrlsim_attention_mask_table(int Lq, int Lk, int A, int rep_keys, int rep_pos0, int nkt, void* tbl, hipStream_t st) {
  if (!tbl) return CTRLSIM_EINVAL;
  const AttnClassHost c{1, Lq, Lk, A, rep_keys, 1, rep_pos0, nkt, 0, 0, 0, 0, 0, 0, nullptr, tbl};
  return launch_attn_mask_tables(1, &c, st);
}
int ctrlsim_attention_tbl(const float* Q, int ldq, int64_t qbs, const void* img, int nkt, float* O, int ldo, int64_t obs, int B, int Lq,
                          int Lk, int A, int rep_keys, int rep_mult, const void* mask_tbl, hipStream_t st) {
  if (!mask_tbl) return CTRLSIM_EINVAL;
  return launch_attention_bf16x6_pre(1, Q, ldq, (long)qbs, img, nkt, O, ldo, (long)obs, nullptr, nullptr, B, Lq, Lk, A, rep_keys,
                                     rep_mult, Lk, mask_tbl, st);
}
int ctrlsim_attention(int mode, const float* Q, int ldq, int64_t qbs, const float* K, const float* V, int ldkv, int64_t kbs,
                      float* O, int ldo, int64_t obs, const int* q_pos, const uint8_t* key_pad, int B, int Lq, int Lk, int A,
                      hipStream_t st) {
  return launch_attention(mode, Q, ldq, (long)qbs, K, V, ldkv, (long)kbs, O, ldo, (long)obs, q_pos, key_pad, B, Lq, Lk, A, st);
}
int ctrlsim_sim_init(int S, int N, int E, const float* init_pose, const float* size, const float* edges, const uint8_t* exists,
                     float* phys, float* hist_states, uint8_t* coll, int Tmax1, float* contact_state, hipStream_t st) {
  return launch_sim_init(S, N, E, init_pose, size, edges, exists, phys, hist_states, coll, Tmax1, contact_state, st);
}
int ctrlsim_sim_set_position(int S, int N, const float* xy, float* phys, hipStream_t st) {
  return launch_sim_set_position(S, N, xy, phys, st);
}
int64_t ctrlsim_sim_contact_floats(int N) { return N < 1 ? 0 : (int64_t)N * (N - 1) / 2 * 20 + 6 + 28 * (int64_t)N; }
int ctrlsim_sim_step(int S, int N, int E, const int* act_tok, const double* act_f64, const double* disc6, const float* size,
                     const float* edges, const uint8_t* exists, float* phys, float* hist_states, uint8_t* coll,
                     double* applied, int t, int Tmax1, float dt, int mode, float* contact_state, hipStream_t st) {
  if (!disc6) return CTRLSIM_EINVAL;
  return launch_sim_step(S, N, E, act_tok, act_f64, disc6, size, edges, exists, phys, hist_states, coll, applied, t, Tmax1, dt,
                         mode, contact_state, nullptr, st);
}
int ctrlsim_sim_step_expert(int S, int N, int E, const int* act_tok, const double* act_f64, const double* disc6, const float* size,
                            const float* edges, const uint8_t* exists, float* phys, float* hist_states, uint8_t* coll,
                            double* applied, int t, int Tmax1, float dt, float* contact_state, const float* expert, hipStream_t st) {
  if (!disc6) return CTRLSIM_EINVAL;
  return launch_sim_step(S, N, E, act_tok, act_f64, disc6, size, edges, exists, phys, hist_states, coll, applied, t, Tmax1, dt, 0,
                         contact_state, expert, st);
}
int ctrlsim_group_build(int S, int N, int A, int T, int t, int Tmax1, double dist_thresh, const float* hist_states,
                        const int* eval_order, int has_roads, uint64_t* persist, int* n_groups, int* grp_focal,
                        uint64_t* grp_ids, uint64_t* grp_members, int* own_g, int* mem_g, uint8_t* tilted, hipStream_t st) {
  return launch_group_build(S, N, A, T, t, Tmax1, dist_thresh, hist_states, eval_order, has_roads,
                            (unsigned long long*)persist, n_groups, grp_focal, (unsigned long long*)grp_ids,
                            (unsigned long long*)grp_members, own_g, mem_g, tilted, st);
}
int ctrlsim_ctx_index(int s0, int s1, int N, const int* n_groups, const int* grp_focal, const uint64_t* grp_ids,
                      const int* own_g, const int* mem_g, int* ctx_scn, int* ctx_grp, int* own_ctx, int* own_slot,
                      int* mem_ctx, int* mem_slot, int* ctx_base, hipStream_t st) {
  return launch_ctx_index(s0, s1, N, n_groups, grp_focal, (const unsigned long long*)grp_ids, own_g, mem_g, ctx_scn, ctx_grp,
                          own_ctx, own_slot, mem_ctx, mem_slot, ctx_base, st);
}
int ctrlsim_groups_changed(int S, int N, const int* n_groups, const int* grp_focal, const uint64_t* grp_ids, const int* ref_n,
                           const int* ref_focal, const uint64_t* ref_ids, int* flag, hipStream_t st) {
  return launch_groups_changed(S, N, n_groups, grp_focal, (const unsigned long long*)grp_ids, ref_n, ref_focal,
                               (const unsigned long long*)ref_ids, flag, st);
}
int ctrlsim_build_context(int B, int N, int A, int T, int t, int Tq, int tt_first, int Tmax1, int Tmax, int P_all, int P, int NP,
                          const int* ctx_scn, const int* ctx_grp, const int* grp_focal, const uint64_t* grp_ids,
                          const float* hist_states, const int* hist_tok, const int* hist_rtg, const double* goals,
                          const float* types, const float* roads, const float* road_types, const int* zero4,
                          const ctrlsim_ctx* out, hipStream_t st) {
  if (!out || !zero4) return CTRLSIM_EINVAL;
  CtxOut o{out->st12, out->exist, out->goal5, out->act_tok, out->rtg_bin, out->tstep, out->slot_gid, out->road_pts,
           out->road_types};
  return launch_build_context(B, N, A, T, t, Tq, tt_first, Tmax1, Tmax, P_all, P, NP, ctx_scn, ctx_grp, grp_focal,
                              (const unsigned long long*)grp_ids, hist_states, hist_tok, hist_rtg, goals, types, roads,
                              road_types, zero4, o, st);
}
int ctrlsim_build_context_c(int n, const int* B, const int* A, const ctrlsim_ctx* out, int N, int T, int t, int Tq, int tt_first,
                            int Tmax1, int Tmax, int P_all, int P, int NP, const int* ctx_scn, const int* ctx_grp,
                            const int* grp_focal, const uint64_t* grp_ids, const float* hist_states, const int* hist_tok,
                            const int* hist_rtg, const double* goals, const float* types, const float* roads,
                            const float* road_types, const int* zero4, hipStream_t st) {
  if (!out || !zero4 || !B || !A || n < 1 || n > MAXC) return CTRLSIM_EINVAL;
  CtxOut o[MAXC];
  for (int k = 0; k < n; ++k)
    o[k] = CtxOut{out[k].st12, out[k].exist, out[k].goal5, out[k].act_tok, out[k].rtg_bin, out[k].tstep, out[k].slot_gid,
                  out[k].road_pts, out[k].road_types};
  return launch_build_context_classes(n, B, A, o, N, T, t, Tq, tt_first, Tmax1, Tmax, P_all, P, NP, ctx_scn, ctx_grp, grp_focal,
                                      (const unsigned long long*)grp_ids, hist_states, hist_tok, hist_rtg, goals, types, roads,
                                      road_types, zero4, st);
}
int ctrlsim_sample_rtg(const float* rtg_logits, int A, int R, const int* own_ctx, const int* own_slot, const uint8_t* tilted,
                       const double* tilt3, const double* tilt_scn, const float* noise, uint64_t seed, const int64_t* scenario_id,
                       int t, int* hist_rtg, int S, int N, int Tmax, hipStream_t st) {
  if (!tilt3) return CTRLSIM_EINVAL;
  return launch_sample_rtg(rtg_logits, A, R, own_ctx, own_slot, nullptr, tilted, tilt3, tilt_scn, noise, seed, scenario_id, t, hist_rtg,
                           S, N, Tmax, st);
}
int ctrlsim_sample_rtg_rows(const float* rtg_logits, const int* ctx_row0, int R, const int* own_ctx, const int* own_slot,
                            const uint8_t* tilted, const double* tilt3, const double* tilt_scn, const float* noise, uint64_t seed,
                            const int64_t* scenario_id, int t, int* hist_rtg, int S, int N, int Tmax, hipStream_t st) {
  if (!tilt3 || !ctx_row0) return CTRLSIM_EINVAL;
  return launch_sample_rtg(rtg_logits, 0, R, own_ctx, own_slot, ctx_row0, tilted, tilt3, tilt_scn, noise, seed, scenario_id, t,
                           hist_rtg, S, N, Tmax, st);
}
int ctrlsim_sample_action_rows(const float* act_logits, const int* ctx_row0, int V, const int* mem_ctx, const int* mem_slot,
                               float temperature, double top_p, const float* noise, uint64_t seed, const int64_t* scenario_id,
                               int t, int* hist_tok, int* act_now, int S, int N, int Tmax, int zero_token, hipStream_t st) {
  if (!ctx_row0) return CTRLSIM_EINVAL;
  return launch_sample_action(act_logits, 0, V, mem_ctx, mem_slot, ctx_row0, temperature, top_p, noise, seed, scenario_id, t,
                              hist_tok, act_now, S, N, Tmax, zero_token, st);
}
int ctrlsim_group_size_hist(int S, int N, const int* n_groups, const uint64_t* grp_ids, int nb, const int* sizes, int* hist,
                            hipStream_t st) {
  return launch_group_size_hist(S, N, n_groups, (const unsigned long long*)grp_ids, nb, sizes, hist, st);
}
int ctrlsim_ctx_index_classes(int s0, int s1, int N, int A, const int* n_groups, const uint64_t* grp_ids, const int* own_g,
                              const int* mem_g, int nb, const int* sizes, int* ctx_scn, int* ctx_grp, int* ctx_row0,
                              int* ctx_of_group, int* own_ctx, int* own_slot, int* mem_ctx, int* mem_slot, hipStream_t st) {
  return launch_ctx_index_classes(s0, s1, N, A, n_groups, (const unsigned long long*)grp_ids, own_g, mem_g, nb, sizes, ctx_scn,
                                  ctx_grp, ctx_row0, ctx_of_group, own_ctx, own_slot, mem_ctx, mem_slot, st);
}
int ctrlsim_sample_action(const float* act_logits, int A, int V, const int* mem_ctx, const int* mem_slot, float temperature,
                          double top_p, const float* noise, uint64_t seed, const int64_t* scenario_id, int t, int* hist_tok,
                          int* act_now, int S, int N, int Tmax, int zero_token, hipStream_t st) {
  return launch_sample_action(act_logits, A, V, mem_ctx, mem_slot, nullptr, temperature, top_p, noise, seed, scenario_id, t, hist_tok,
                              act_now, S, N, Tmax, zero_token, st);
}

}  // extern "C"
