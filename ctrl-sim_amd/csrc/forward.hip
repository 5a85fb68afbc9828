// Host-side orchestration of the CtRL-Sim forward for a batch of B agent-local contexts (gfx950).
//
// Replaces, for the rollout, `CtRLSim.forward` = Decoder(Encoder(data)) (models/ctrl_sim.py:41-45;
// modules/encoder.py:50-178, modules/map_encoder.py:34-53, modules/decoder.py:39-79) as it is called twice per
// focal group and step by AutoregressivePolicy.predict (policies/autoregressive_policy.py:189-210):
//   pass 1  -> RTG logits of the current-timestep STATE tokens      (predict_rtg head,    decoder.py:74-77)
//   pass 2  -> action logits of the current-timestep RTG tokens     (predict_action head, decoder.py:58-62)
// Exact (in real arithmetic) savings over the reference's two dense passes:
//   * only the first Tq = token_index+1 window steps are materialised (later tokens are invisible to the queries);
//   * heads are evaluated for the A current-timestep tokens only, and the last decoder layer computes K/V for all
//     tokens but attention/FFN only for those A queries;
//   * pass 2 re-evaluates nothing but the A RTG tokens of the current timestep through the 4 layers against the
//     cached per-layer K/V of pass 1 (the sampled RTG only changes tokens that are visible to the same agent's
//     RTG/action tokens of that timestep: SURVEY.md §8a M6 corollary);
//   * map encoder / embedding linear chains are folded at pack time (map_encoder.hip, embed.hip).
// Everything is fp32; GEMMs and attention run on the f32-input MFMA.  No allocation, no synchronisation: the caller
// provides the workspace (ctrlsim_forward_workspace_bytes) and a stream.
#include <string>
#include <unordered_map>
#include <vector>

#include "split.h"
#include "../../include/ctrlsim.h"

// launchers from the other translation units
int launch_gemm_nt(const float*, int, const float*, int, const float*, const float*, int, float*, int, int, int, int, int,
                   hipStream_t);
int launch_gemm_nt_bf16x6(const float*, int, const void*, int, int, const float*, const float*, int, float*, int, int, int,
                          int, int, const float*, const float*, hipStream_t);
int launch_layernorm256(const float*, int, const float*, int, const float*, const float*, float*, int, int, int,
                        hipStream_t);
int launch_gemm_nt_bf16x6_kv(const float*, int, const void*, int, int, const float*, const float*, int, float*, int, int, int,
                             int, int, const float*, const float*, void*, int, int, int, int, int, hipStream_t);
int launch_kv_zero_tail(int, int, int, int, void*, hipStream_t);
int launch_ffn_fused_bf16x6(const float*, int, const void*, const float*, const void*, const float*, const float*, const float*,
                            float*, int, int, int, hipStream_t);
int launch_kv_split(const float*, const float*, int, long, int, int, int, void*, hipStream_t);
int launch_kv_split_rows(const float*, const float*, int, long, const int*, int, int, int, void*, hipStream_t);
int launch_attention_bf16x6_pre(int, const float*, int, long, const void*, int, float*, int, long, const int*,
                                const unsigned char*, int, int, int, int, int, int, int, hipStream_t);
int launch_in_mlp(const float*, int, int, const float*, const float*, const float*, const float*, float*, int, int,
                  hipStream_t);
int launch_row_copy(const float*, int, float*, int, const int*, int, int, int, hipStream_t);
int launch_attention(int, const float*, int, long, const float*, const float*, int, long, float*, int, long, const int*,
                     const unsigned char*, int, int, int, int, hipStream_t);
struct EmbedTables { const float *act, *rtg_g, *rtg_v, *rtg_r, *rtg_bias, *tstep, *agent, *ln_g, *ln_b; int rtg_linear; };
int launch_assemble_tokens(int, int, int, int, const float*, const float*, const float*, const int*, const int*, const int*,
                           EmbedTables, float*, float*, int, int, unsigned char*, hipStream_t);
int launch_assemble_rows(int, int, int, int, int, const int*, const float*, const float*, const float*, const int*, const int*,
                         const int*, EmbedTables, float*, hipStream_t);
int launch_assemble_rtg_rows(int, int, int, int, int, int, int, int, const int*, const int*, const int*, const float*,
                             const int*, EmbedTables, const int*, float*, hipStream_t);
struct MapPoolWeights { const float *Wc, *G, *ln_b, *U, *cb, *Mt, *mb; };
int launch_map_pool(int, int, int, int, const float*, MapPoolWeights, float*, unsigned char*, hipStream_t);

namespace {

struct Lin { const float* w; const float* b; const void* w3 = nullptr; int ntot = 0; int n0 = 0; };
struct LNp { const float* g; const float* b; };
struct Mlp { Lin l0; LNp ln; Lin l3; };
struct EncLayer { Lin qkv, out, lin1, lin2; LNp n1, n2; const void *w1p = nullptr, *w2p = nullptr; };
struct DecLayer { Lin qkv, out, cq, ckv, cout, lin1, lin2; LNp n1, n2, n3; const void *w1p = nullptr, *w2p = nullptr; };

}  // namespace

struct ctrlsim_model {
  ctrlsim_dims d;
  // embeddings
  Mlp embed_state, embed_goal;             // only l0 + ln used (l3 folded)
  Lin fold_state, fold_goal;
  EmbedTables tb;
  // map encoder
  MapPoolWeights mp;
  Lin map_out;
  LNp map_n1, map_n2;
  Mlp map_feats, road_type, road_fuse;
  std::vector<EncLayer> enc;
  std::vector<DecLayer> dec;
  Mlp head_action, head_rtg;
  int zero_rtg[3];
};

#define CHK(x)            \
  do {                    \
    int _e = (x);         \
    if (_e != 0) return _e; \
  } while (0)

extern "C" int ctrlsim_model_create(const ctrlsim_dims* dims, const float* dev_weights, int n, const char* const* names,
                                    const int64_t* offsets, ctrlsim_model** out) {
  if (!dims || !dev_weights || !names || !offsets || !out) return CTRLSIM_EINVAL;
  if (dims->D != DM || dims->H != NHEAD || dims->A < 1 || dims->A > 64 || dims->variant < 0 || dims->variant > 3)
    return CTRLSIM_EINVAL;
  std::unordered_map<std::string, const float*> tab;
  for (int i = 0; i < n; ++i) tab[names[i]] = dev_weights + offsets[i];
  bool ok = true;
  auto P = [&](const std::string& k) -> const float* {
    auto it = tab.find(k);
    if (it == tab.end()) { ok = false; return nullptr; }
    return it->second;
  };
  auto P3 = [&](const std::string& k) -> const void* {     // optional bf16x3 planes
    auto it = tab.find(k + "#bf3");
    return it == tab.end() ? nullptr : static_cast<const void*>(it->second);
  };
  auto PX = [&](const std::string& k) -> const void* {     // optional extra operand images
    auto it = tab.find(k);
    return it == tab.end() ? nullptr : static_cast<const void*>(it->second);
  };
  auto lin = [&](const std::string& k) { return Lin{P(k + ".weight"), P(k + ".bias"), P3(k + ".weight"), 0, 0}; };
  auto lnp = [&](const std::string& k) { return LNp{P(k + ".weight"), P(k + ".bias")}; };
  auto mlp = [&](const std::string& k) { return Mlp{lin(k + ".mlp.0"), lnp(k + ".mlp.1"), lin(k + ".mlp.3")}; };
  ctrlsim_model* m = new ctrlsim_model();
  m->d = *dims;
  m->embed_state = mlp("encoder.embed_state");
  m->embed_goal = mlp("encoder.embed_goal");
  m->fold_state = Lin{P("fold.embed_state.w"), nullptr, P3("fold.embed_state.w"), 0, 0};
  m->fold_goal = Lin{P("fold.embed_goal.w"), P("fold.embed_goal.b"), P3("fold.embed_goal.w"), 0, 0};
  m->tb = EmbedTables{P("encoder.embed_action.weight"), P("fold.rtg_table_goal"), P("fold.rtg_table_veh"),
                      P("fold.rtg_table_road"), P("fold.rtg_bias"), P("encoder.embed_timestep.weight"),
                      P("encoder.embed_agent_id.weight"), P("encoder.embed_ln.weight"), P("encoder.embed_ln.bias"),
                      dims->variant == 3 ? 1 : 0};
  const std::string me = "encoder.map_encoder.";
  m->mp = MapPoolWeights{P("fold.map.Wc"), P("fold.map.G"), P(me + "road_pts_encoder.mlp.1.bias"),
                         P("fold.map.U"), P("fold.map.cb"), P("fold.map.Mt"), P("fold.map.mb")};
  m->map_out = lin(me + "road_pts_attn_layer.out_proj");
  m->map_n1 = lnp(me + "norm1");
  m->map_n2 = lnp(me + "norm2");
  m->map_feats = mlp(me + "map_feats");
  m->road_type = mlp(me + "road_type_encoder");
  m->road_fuse = mlp(me + "road_road_type_encoder");
  for (int i = 0; i < dims->NE; ++i) {
    const std::string p = "encoder.transformer_encoder.layers." + std::to_string(i);
    EncLayer L;
    L.qkv = Lin{P(p + ".self_attn.in_proj_weight"), P(p + ".self_attn.in_proj_bias"), P3(p + ".self_attn.in_proj_weight"), 3 * DM, 0};
    L.out = lin(p + ".self_attn.out_proj");
    L.lin1 = lin(p + ".linear1");
    L.lin2 = lin(p + ".linear2");
    L.n1 = lnp(p + ".norm1");
    L.n2 = lnp(p + ".norm2");
    L.w1p = PX(p + ".ffn#w1p");
    L.w2p = PX(p + ".ffn#w2p");
    m->enc.push_back(L);
  }
  for (int i = 0; i < dims->ND; ++i) {
    const std::string p = "decoder.transformer_decoder.layers." + std::to_string(i);
    DecLayer L;
    L.qkv = Lin{P(p + ".self_attn.in_proj_weight"), P(p + ".self_attn.in_proj_bias"), P3(p + ".self_attn.in_proj_weight"), 3 * DM, 0};
    L.out = lin(p + ".self_attn.out_proj");
    const float* cw = P(p + ".multihead_attn.in_proj_weight");
    const float* cb = P(p + ".multihead_attn.in_proj_bias");
    const void* cw3 = P3(p + ".multihead_attn.in_proj_weight");
    L.cq = Lin{cw, cb, cw3, 3 * DM, 0};
    L.ckv = Lin{cw ? cw + DM * DM : nullptr, cb ? cb + DM : nullptr, cw3, 3 * DM, DM};
    L.cout = lin(p + ".multihead_attn.out_proj");
    L.lin1 = lin(p + ".linear1");
    L.lin2 = lin(p + ".linear2");
    L.n1 = lnp(p + ".norm1");
    L.n2 = lnp(p + ".norm2");
    L.n3 = lnp(p + ".norm3");
    L.w1p = PX(p + ".ffn#w1p");
    L.w2p = PX(p + ".ffn#w2p");
    m->dec.push_back(L);
  }
  m->head_action = mlp("decoder.predict_action");
  if (dims->variant == 0) m->head_rtg = mlp("decoder.predict_rtg");      // the IL / Trajeglish models have no such head
  m->zero_rtg[0] = 0; m->zero_rtg[1] = 35; m->zero_rtg[2] = 35;
  if (!ok) { delete m; return CTRLSIM_EINVAL; }
  *out = m;
  return CTRLSIM_OK;
}

extern "C" void ctrlsim_model_destroy(ctrlsim_model* m) { delete m; }

// ------------------------------------------------------------------------------------------------ workspace
namespace {
// Shape of a uniform batch of contexts.  Token rows of agent slots that do not exist anywhere in the window are all equal
// (every embedding is multiplied by the existence flag before embed_ln, modules/encoder.py:127-133) and stay equal through the
// decoder (no key padding on the targets; the structured mask treats all of them alike), so a context with n < A vehicles
// can be evaluated with Actx >= n + 1 slots: Areg = Actx - 1 regular slots, and ONE representative slot standing for the
// mult = A - Areg padded slots of the reference's 24-slot layout (attention_bf16x6.hip: multiplicity of its keys).  A context's
// L token rows: regular (tt, a < Areg, k) at (tt*Areg + a)*3 + k, representative (tt, k) at Lreg + 3*tt + k; its keys sit in
// the tiles from key rep_k0 = 64*ceil(Lreg/64) on.  Actx == A: the plain layout (rep = 0).  Exact in real arithmetic; the engine
// sorts the contexts of a step into a few such shapes (engine.py).
struct Shape {
  int A, Areg, rep, mult;
  int rows(int Tq) const { return Tq * 3 * (Areg + rep); }
  int lreg(int Tq) const { return Tq * 3 * Areg; }
  int rep_k0(int Tq) const { return (lreg(Tq) + 63) / 64 * 64; }
  int nkt(int Tq) const { return (lreg(Tq) + 63) / 64 + rep * ((3 * Tq + 63) / 64); }
};
Shape shape_of(const ctrlsim_dims& d, int Actx) {
  Shape s;
  s.A = Actx; s.rep = Actx < d.A ? 1 : 0; s.Areg = Actx - s.rep; s.mult = d.A - s.Areg;
  return s;
}

struct Ws {
  float *hS, *S2, *hG, *Gp, *X, *src, *attn_pre, *m1, *m2, *cat, *tfh, *eqkv, *eatt, *etmp, *effn;
  float *memkv[8], *qkv[8];
  float *att, *tmp, *qc, *ffn;
  float *xc, *xc2, *tmpc, *attc, *qkvc, *qcc, *ffnc, *headh;
  unsigned char* src_pad;
  int *pos_state, *pos_rtg, *idx_state, *idx_rtg, *idx_poly, *key_all;
  // cached incremental path: up to 4A new rows per context
  float *xn, *tmpn, *attn_n, *qkvn, *qcn, *ffnn;
  int *pos_new, *key_new, *src_new, *idx_new, *idx_state_in_new;
  // split-bf16 K/V images (attention_bf16x6.hip: kv_split_kernel): decoder self-attention per layer, memory K/V per
  // layer, scene encoder (reused by its layers); nkt = 64-key tiles per context
  void *img_dec[8], *img_mem[8], *img_enc;
  int nkt_dec, nkt_mem;
  size_t img_dec_bytes;
  size_t bytes;
};

Ws carve(const ctrlsim_dims& d, const Shape& sh, int B, int Tq, char* base) {
  Ws w;
  size_t off = 0;
  auto take = [&](size_t nbytes) -> char* {
    char* p = base ? base + off : nullptr;
    off += (nbytes + 255) & ~size_t(255);
    return p;
  };
  const size_t L = (size_t)sh.rows(Tq), M = (size_t)d.P + sh.A;
  const size_t rL = B * L, rM = B * M, rA = (size_t)B * sh.A, rS = (size_t)B * Tq * sh.A, rP = (size_t)B * d.P;
  auto F = [&](size_t rows, size_t cols) { return reinterpret_cast<float*>(take(rows * cols * sizeof(float))); };
  w.hS = F(rS, DM); w.S2 = F(rS, DM); w.hG = F(rA, DM); w.Gp = F(rA, DM);
  w.X = F(rL, DM); w.src = F(rM, DM);
  w.attn_pre = F(rP, DM); w.m1 = F(rP, DM); w.m2 = F(rP, DM); w.cat = F(rP, 2 * DM); w.tfh = F(rP, DM);
  w.eqkv = F(rM, 3 * DM); w.eatt = F(rM, DM); w.etmp = F(rM, DM); w.effn = F(rM, d.F);
  for (int i = 0; i < d.ND; ++i) w.memkv[i] = F(rM, 2 * DM);
  for (int i = 0; i < d.ND; ++i) w.qkv[i] = F(rL, 3 * DM);
  w.att = F(rL, DM); w.tmp = F(rL, DM); w.qc = F(rL, DM); w.ffn = F(rL, d.F);
  w.xc = F(rA, DM); w.xc2 = F(rA, DM); w.tmpc = F(rA, DM); w.attc = F(rA, DM); w.qkvc = F(rA, 3 * DM);
  w.qcc = F(rA, DM); w.ffnc = F(rA, d.F); w.headh = F(rA, DM);
  w.src_pad = reinterpret_cast<unsigned char*>(take(rM));
  w.pos_state = reinterpret_cast<int*>(take(sh.A * sizeof(int)));
  w.pos_rtg = reinterpret_cast<int*>(take(sh.A * sizeof(int)));
  w.idx_state = reinterpret_cast<int*>(take(rA * sizeof(int)));
  w.idx_rtg = reinterpret_cast<int*>(take(rA * sizeof(int)));
  w.idx_poly = reinterpret_cast<int*>(take(rP * sizeof(int)));
  w.key_all = reinterpret_cast<int*>(take(L * sizeof(int)));
  const size_t rN = rA * 4;
  w.xn = F(rN, DM); w.tmpn = F(rN, DM); w.attn_n = F(rN, DM); w.qkvn = F(rN, 3 * DM); w.qcn = F(rN, DM); w.ffnn = F(rN, d.F);
  w.pos_new = reinterpret_cast<int*>(take(4 * sh.A * sizeof(int)));
  w.key_new = reinterpret_cast<int*>(take(4 * sh.A * sizeof(int)));
  w.src_new = reinterpret_cast<int*>(take(4 * sh.A * sizeof(int)));
  w.idx_new = reinterpret_cast<int*>(take(rN * sizeof(int)));
  w.idx_state_in_new = reinterpret_cast<int*>(take(rA * sizeof(int)));
  w.nkt_dec = sh.nkt(Tq);
  w.nkt_mem = (int)((M + 63) / 64);
  const size_t tile_bytes = (size_t)2 * NPL * 64 * HD * 2;   // 8 KB per plane pair of a (context, head, tile) image
  w.img_dec_bytes = (size_t)B * NHEAD * w.nkt_dec * tile_bytes;
  for (int i = 0; i < d.ND; ++i) w.img_dec[i] = take(w.img_dec_bytes);
  for (int i = 0; i < d.ND; ++i) w.img_mem[i] = take((size_t)B * NHEAD * w.nkt_mem * tile_bytes);
  w.img_enc = take((size_t)B * NHEAD * w.nkt_mem * tile_bytes);
  w.bytes = off;
  return w;
}

// qoff: token type whose rows feed the first pass's head (0 = state tokens; 2 = action tokens, Trajeglish).  Index lists of the
// Areg current-step query rows per context (positions in the context's row order; regular rows: position == key), the
// polyline rows of the scene-encoder source, and key_all: row -> key position in the K/V images (Tq window steps).
__global__ void fill_index_kernel(int B, int Areg, int L, int Lreg, int rep_k0, int ti, int P, int M, int qoff, int* pos_state,
                                  int* pos_rtg, int* idx_state, int* idx_rtg, int* idx_poly, int* key_all) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * P) idx_poly[i] = (i / P) * M + (i % P);
  if (i < Areg) { pos_state[i] = (ti * Areg + i) * 3 + qoff; pos_rtg[i] = (ti * Areg + i) * 3 + 1; }
  if (i < B * Areg) {
    const int b = i / Areg, a = i - b * Areg;
    idx_state[i] = b * L + (ti * Areg + a) * 3 + qoff;
    idx_rtg[i] = b * L + (ti * Areg + a) * 3 + 1;
  }
  if (i < L) key_all[i] = i < Lreg ? i : rep_k0 + (i - Lreg);
}

// y = act(x W^T + b [+ R]) through the bf16x6 MFMA kernel when the packed planes exist, else the f32-input MFMA kernel
int gemm(const Lin& L, const float* x, int ldx, const float* R, int ldr, float* y, int ldy, int rows, int n, int k, int relu,
         hipStream_t st) {
  if (L.w3 && k % 32 == 0 && ctrlsim_option(OPT_GEMM_IMPL) == 1)
    return launch_gemm_nt_bf16x6(x, ldx, L.w3, L.ntot ? L.ntot : n, L.n0, L.b, R, ldr, y, ldy, rows, n, k, relu, nullptr,
                                 nullptr, st);
  return launch_gemm_nt(x, ldx, L.w, k, L.b, R, ldr, y, ldy, rows, n, k, relu, st);
}

// y = [relu] LayerNorm(x W^T + b [+ R]) for the 256-wide blocks: one kernel on the bf16x6 path (LN in the GEMM epilogue;
// y may alias R), GEMM into `tmp` + layernorm256 on the f32-input path
int gemm_ln(const Lin& L, const LNp& n, const float* x, int ldx, const float* R, int ldr, float* y, int ldy, float* tmp,
            int rows, int k, int relu, hipStream_t st) {
  if (L.w3 && k % 32 == 0 && ctrlsim_option(OPT_GEMM_IMPL) == 1)
    return launch_gemm_nt_bf16x6(x, ldx, L.w3, L.ntot ? L.ntot : DM, L.n0, L.b, R, ldr, y, ldy, rows, DM, k, relu, n.g, n.b, st);
  CHK(launch_gemm_nt(x, ldx, L.w, k, L.b, R, ldr, tmp, DM, rows, DM, k, 0, st));
  return launch_layernorm256(tmp, DM, nullptr, 0, n.g, n.b, y, ldy, rows, relu, st);
}

// x <- LayerNorm(x + linear2(relu(linear1(x)))): one fused kernel (hidden tile in registers) or Linear + Linear/LN
int ffn_block(const Lin& l1, const Lin& l2, const LNp& n, const void* w1p, const void* w2p, float* x, float* hidden, float* tmp,
              int rows, int F, hipStream_t st) {
  if (w1p && w2p && !(F & 31) && ctrlsim_option(OPT_FFN_FUSED) >= 1 && ctrlsim_option(OPT_GEMM_IMPL) == 1)
    return launch_ffn_fused_bf16x6(x, DM, w1p, l1.b, w2p, l2.b, n.g, n.b, x, DM, rows, F, st);
  CHK(gemm(l1, x, DM, nullptr, 0, hidden, F, rows, F, DM, 1, st));
  return gemm_ln(l2, n, hidden, F, x, DM, x, DM, tmp, rows, F, 0, st);
}

// Attention over K/V given both as fp32 rows and (when the split-bf16 path is selected) as pre-split images.
// Rep: the representative-token region of a compact context (split-operand path only; rep_keys == 0: none).
struct Rep { int keys = 0, mult = 1, pos0 = 0; };
inline bool presplit() { return ctrlsim_option(OPT_ATTN_IMPL) == 1; }
int attention_kv(int mode, const float* Q, int ldq, long qbs, const float* K, const float* V, int ldkv, long kbs,
                 const void* img, int nkt, float* O, int ldo, long obs, const int* q_pos, const unsigned char* key_pad, int B,
                 int Lq, int Lk, int A, hipStream_t st, Rep rep = Rep()) {
  if (presplit())
    return launch_attention_bf16x6_pre(mode, Q, ldq, qbs, img, nkt, O, ldo, obs, q_pos, key_pad, B, Lq, Lk, A, rep.keys, rep.mult,
                                       rep.pos0, st);
  if (rep.keys) return CTRLSIM_EINVAL;
  return launch_attention(mode, Q, ldq, qbs, K, V, ldkv, kbs, O, ldo, obs, q_pos, key_pad, B, Lq, Lk, A, st);
}
int kv_split(const float* K, const float* V, int ldkv, long kbs, int B, int Lk, int nkt, void* img, hipStream_t st) {
  return presplit() ? launch_kv_split(K, V, ldkv, kbs, B, Lk, nkt, img, st) : 0;
}
int kv_split_rows(const float* K, const float* V, int ldkv, long kbs, const int* pos, int B, int R, int nkt, void* img,
                  hipStream_t st) {
  return presplit() ? launch_kv_split_rows(K, V, ldkv, kbs, pos, B, R, nkt, img, st) : 0;
}
// Linear whose last 512 output columns are attention keys / values of B contexts x Lk rows: y[:, :kcol0] as fp32 rows,
// K / V as split images straight from the GEMM epilogue when both bf16x6 kernels are selected (the fp32 K / V columns
// of y are then NOT written); otherwise GEMM + kv_split.  Rows >= Lreg of a context go to the keys from rep_k0 on
// (compact contexts; Lreg == Lk: plain layout); key_all = the row -> key list of the fallback.
int gemm_kv(const Lin& L, const float* x, int ldx, float* y, int ldy, int B, int Lk, int n, int k, int kcol0, void* img, int nkt,
            hipStream_t st, int Lreg = 0, int rep_k0 = 0, const int* key_all = nullptr, size_t img_bytes = 0) {
  const int rows = B * Lk;
  if (Lreg <= 0 || Lreg >= Lk) { Lreg = Lk; rep_k0 = 0; }
  if (presplit() && L.w3 && k % 16 == 0 && ctrlsim_option(OPT_GEMM_IMPL) == 1 && !(Lk & 3) && !(Lreg & 3) && Lk >= 32) {
    CHK(launch_kv_zero_tail(B, 0, Lreg, nkt, img, st));
    if (Lreg < Lk) CHK(launch_kv_zero_tail(B, rep_k0, Lk - Lreg, nkt, img, st));
    return launch_gemm_nt_bf16x6_kv(x, ldx, L.w3, L.ntot ? L.ntot : n, L.n0, L.b, nullptr, 0, y, ldy, rows, n, k, 0, nullptr,
                                    nullptr, img, Lk, nkt, kcol0, Lreg, rep_k0, st);
  }
  CHK(gemm(L, x, ldx, nullptr, 0, y, ldy, rows, n, k, 0, st));
  if (Lreg == Lk) return kv_split(y + kcol0, y + kcol0 + DM, ldy, (long)Lk * ldy, B, Lk, nkt, img, st);
  if (!presplit()) return CTRLSIM_EINVAL;
  if (hipMemsetAsync(img, 0, img_bytes, st) != hipSuccess) return CTRLSIM_ELAUNCH;
  return kv_split_rows(y + kcol0, y + kcol0 + DM, ldy, (long)Lk * ldy, key_all, B, Lk, nkt, img, st);
}

// index lists of the cached path: the new rows of step t are the action tokens of step t-1 (t > 0) and the three tokens of step
// t, of the Areg regular slots and then of the representative; cache rows live at b * Lf + position with Lf = rows of the
// full window.  pos_new = position in the row order (= the query position the mask uses), key_new = key in the K/V images,
// src_new = ((tt - tt_first) * Actx + slot) * 3 + k in the context tensors.
__global__ void fill_index_cached_kernel(int B, int Actx, int Areg, int Lf, int Lreg_f, int rep_k0, int t, int Rn, int* pos_new,
                                         int* key_new, int* src_new, int* idx_new, int* idx_state_in_new, int* pos_rtg,
                                         int* idx_rtg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int off = t > 0 ? Actx : 0;
  // entry j -> (window step tt, slot a, token k)
  auto decode = [&](int j, int& tt, int& a, int& k) {
    if (t > 0 && j < Actx) { tt = t - 1; a = j; k = 2; return; }
    const int r = j - off;
    tt = t;
    if (r < 3 * Areg) { a = r / 3; k = r - 3 * a; } else { a = Areg; k = r - 3 * Areg; }
  };
  auto pos_of = [&](int tt, int a, int k) { return a < Areg ? (tt * Areg + a) * 3 + k : Lreg_f + 3 * tt + k; };
  if (i < Rn) {
    int tt, a, k;
    decode(i, tt, a, k);
    const int p = pos_of(tt, a, k);
    pos_new[i] = p;
    key_new[i] = a < Areg ? p : rep_k0 + 3 * tt + k;
    src_new[i] = ((tt - (t > 0 ? t - 1 : 0)) * Actx + a) * 3 + k;
  }
  if (i < Areg) pos_rtg[i] = (t * Areg + i) * 3 + 1;
  if (i < B * Rn) {
    int tt, a, k;
    decode(i % Rn, tt, a, k);
    idx_new[i] = (i / Rn) * Lf + pos_of(tt, a, k);
  }
  if (i < B * Areg) {
    const int b = i / Areg, a = i - b * Areg;
    idx_state_in_new[i] = b * Rn + off + 3 * a;
    idx_rtg[i] = b * Lf + (t * Areg + a) * 3 + 1;
  }
}

int mlp_tail(const Mlp& m, const float* h_in, int rows, float* hid, float* out, int n_out, hipStream_t st) {
  // Linear(256->256) -> LN -> ReLU -> Linear(256->n_out)
  CHK(gemm_ln(m.l0, m.ln, h_in, DM, nullptr, 0, hid, DM, hid, rows, DM, 1, st));
  CHK(gemm(m.l3, hid, DM, nullptr, 0, out, n_out, rows, n_out, DM, 0, st));
  return 0;
}

// cross-attention + FFN sub-blocks shared by the full-row and compact-row paths (post-LN, residual fused in GEMM)
int cross_and_ffn(const ctrlsim_model* m, const Shape& sh, const DecLayer& Ld, int layer, const Ws& w, float* x, float* tmp,
                  float* att, float* qc, float* ffn, int rows, int B, int rows_per_b, hipStream_t st) {
  const ctrlsim_dims& d = m->d;
  const int M = d.P + sh.A;
  CHK(gemm(Ld.cq, x, DM, nullptr, 0, qc, DM, rows, DM, DM, 0, st));
  CHK(attention_kv(0, qc, DM, (long)rows_per_b * DM, w.memkv[layer], w.memkv[layer] + DM, 2 * DM, (long)M * 2 * DM,
                   w.img_mem[layer], w.nkt_mem, att, DM, (long)rows_per_b * DM, nullptr, w.src_pad, B, rows_per_b, M, sh.A, st));
  CHK(gemm_ln(Ld.cout, Ld.n2, att, DM, x, DM, x, DM, tmp, rows, DM, 0, st));
  CHK(ffn_block(Ld.lin1, Ld.lin2, Ld.n3, Ld.w1p, Ld.w2p, x, ffn, tmp, rows, d.F, st));
  return 0;
}
// map encoder + scene encoder + per-layer memory K/V (everything that only depends on the frame of the context)
int scene_side(const ctrlsim_model* m, const Shape& sh, const Ws& w, const ctrlsim_ctx* c, int B, float* dbg_seg_emb,
               hipStream_t st) {
  const ctrlsim_dims& d = m->d;
  const int A = sh.A, P = d.P, M = P + A, rM = B * M, rP = B * P;
  // ---- map encoder (map_encoder.py:34-53): rows of `src` 0..P-1 per context
  CHK(launch_map_pool(B, P, d.NP, M, c->road_pts, m->mp, w.attn_pre, w.src_pad, st));
  CHK(gemm_ln(m->map_out, m->map_n1, w.attn_pre, DM, nullptr, 0, w.m1, DM, w.m1, rP, DM, 0, st));            // emb
  CHK(gemm_ln(m->map_feats.l0, m->map_feats.ln, w.m1, DM, nullptr, 0, w.m2, DM, w.m2, rP, DM, 1, st));
  CHK(gemm_ln(m->map_feats.l3, m->map_n2, w.m2, DM, w.m1, DM, w.cat, 2 * DM, w.attn_pre, rP, DM, 0, st));   // cat[:, :256]
  CHK(launch_in_mlp(c->road_types, 8, 8, m->road_type.l0.w, m->road_type.l0.b, m->road_type.ln.g, m->road_type.ln.b, w.tfh,
                    DM, rP, st));
  CHK(gemm(m->road_type.l3, w.tfh, DM, nullptr, 0, w.cat + DM, 2 * DM, rP, DM, DM, 0, st));                                                                                  // cat[:, 256:]
  CHK(gemm_ln(m->road_fuse.l0, m->road_fuse.ln, w.cat, 2 * DM, nullptr, 0, w.m2, DM, w.m2, rP, 2 * DM, 1, st));
  // final Linear -> compact [B*P,256], then scattered into the scene-encoder source rows [b, 0..P-1]
  CHK(gemm(m->road_fuse.l3, w.m2, DM, nullptr, 0, w.m1, DM, rP, DM, DM, 0, st));
  CHK(launch_row_copy(w.m1, DM, w.src, DM, w.idx_poly, rP, DM, 1, st));
  if (dbg_seg_emb) {
    hipError_t e = hipMemcpyAsync(dbg_seg_emb, w.m1, (size_t)rP * DM * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return CTRLSIM_ELAUNCH;
  }
  // ---- scene encoder (encoder.py:155-168): post-LN layers over [polylines || initial states] with key padding
  for (int i = 0; i < d.NE; ++i) {
    const EncLayer& Le = m->enc[i];
    CHK(gemm_kv(Le.qkv, w.src, DM, w.eqkv, 3 * DM, B, M, 3 * DM, DM, DM, w.img_enc, w.nkt_mem, st));
    CHK(attention_kv(0, w.eqkv, 3 * DM, (long)M * 3 * DM, w.eqkv + DM, w.eqkv + 2 * DM, 3 * DM, (long)M * 3 * DM, w.img_enc,
                     w.nkt_mem, w.eatt, DM, (long)M * DM, nullptr, w.src_pad, B, M, M, A, st));
    CHK(gemm_ln(Le.out, Le.n1, w.eatt, DM, w.src, DM, w.src, DM, w.etmp, rM, DM, 0, st));
    CHK(ffn_block(Le.lin1, Le.lin2, Le.n2, Le.w1p, Le.w2p, w.src, w.effn, w.etmp, rM, d.F, st));
  }
  // memory K/V of every decoder layer (cached for pass 2)
  for (int i = 0; i < d.ND; ++i) {
    CHK(gemm_kv(m->dec[i].ckv, w.src, DM, w.memkv[i], 2 * DM, B, M, 2 * DM, DM, 0, w.img_mem[i], w.nkt_mem, st));
  }
  return 0;
}

bool actx_ok(const ctrlsim_dims& d, int Actx) { return Actx >= 2 && Actx <= d.A && (Actx == d.A || (d.variant == 0 && presplit())); }

}  // namespace

extern "C" int64_t ctrlsim_forward_workspace_bytes_a(const ctrlsim_dims* d, int B, int Tq, int Actx) {
  if (!d || B < 1 || Tq < 1 || Tq > d->T || Actx < 2 || Actx > d->A) return CTRLSIM_EINVAL;
  return (int64_t)carve(*d, shape_of(*d, Actx), B, Tq, nullptr).bytes;
}
extern "C" int64_t ctrlsim_forward_workspace_bytes(const ctrlsim_dims* d, int B, int Tq) {
  return d ? ctrlsim_forward_workspace_bytes_a(d, B, Tq, d->A) : CTRLSIM_EINVAL;
}

// ------------------------------------------------------------------------------------------------ pass 1
namespace {
// The full forward over the first Tq window steps; logits of the head that the first pass of the variant needs, for the Areg
// regular slots of every context ([B*Areg, .] rows):
// CtRL-Sim: predict_rtg on the state tokens of the current step; IL: predict_action on the same rows; Trajeglish:
// predict_action on the action tokens (decoder.py:55-77).
int forward_full(const ctrlsim_model* m, int B, int Tq, int Actx, const ctrlsim_ctx* c, void* workspace, float* logits,
                 float* dbg_seg_emb, hipStream_t st) {
  const ctrlsim_dims& d = m->d;
  const int variant = d.variant, amode = 1 + variant, qoff = variant == 2 ? 2 : 0;
  if (variant && !presplit()) return CTRLSIM_EINVAL;       // the IL / Trajeglish masks live in the split-bf16 attention only
  if (!actx_ok(d, Actx)) return CTRLSIM_EINVAL;
  const Shape sh = shape_of(d, Actx);
  const Ws w = carve(d, sh, B, Tq, static_cast<char*>(workspace));
  const int A = sh.A, Ar = sh.Areg, P = d.P, M = P + A, L = sh.rows(Tq), Lreg = sh.lreg(Tq), ti = Tq - 1;
  const int rL = B * L, rA = B * A, rQ = B * Ar, rS = B * Tq * A, rP = B * P;
  const Rep rep{sh.rep * 3 * Tq, sh.mult, Lreg};
  const int nidx = max(max(rA, rP), L);
  hipLaunchKernelGGL(fill_index_kernel, dim3((nidx + 255) / 256), dim3(256), 0, st, B, Ar, L, Lreg, sh.rep_k0(Tq), ti, P, M, qoff,
                     w.pos_state, w.pos_rtg, w.idx_state, w.idx_rtg, w.idx_poly, w.key_all);
  // ---- token embeddings (encoder.py:95-153)
  CHK(launch_in_mlp(c->st12, 12, 12, m->embed_state.l0.w, m->embed_state.l0.b, m->embed_state.ln.g, m->embed_state.ln.b,
                    w.hS, DM, rS, st));
  CHK(gemm(m->fold_state, w.hS, DM, nullptr, 0, w.S2, DM, rS, DM, DM, 0, st));
  CHK(launch_in_mlp(c->goal5, 5, 5, m->embed_goal.l0.w, m->embed_goal.l0.b, m->embed_goal.ln.g, m->embed_goal.ln.b, w.hG,
                    DM, rA, st));
  CHK(gemm(m->fold_goal, w.hG, DM, nullptr, 0, w.Gp, DM, rA, DM, DM, 0, st));
  CHK(launch_assemble_tokens(B, Tq, A, Ar, w.S2, w.Gp, c->exist, c->act_tok, c->rtg_bin, c->tstep, m->tb, w.X, w.src, M, P,
                             w.src_pad, st));
  CHK(scene_side(m, sh, w, c, B, dbg_seg_emb, st));
  // ---- decoder (decoder.py:52): layers 0..ND-2 on all L tokens
  for (int i = 0; i < d.ND; ++i) {
    const DecLayer& Ld = m->dec[i];
    CHK(gemm_kv(Ld.qkv, w.X, DM, w.qkv[i], 3 * DM, B, L, 3 * DM, DM, DM, w.img_dec[i], w.nkt_dec, st, Lreg, sh.rep_k0(Tq),
                w.key_all, w.img_dec_bytes));
    if (i < d.ND - 1) {
      CHK(attention_kv(amode, w.qkv[i], 3 * DM, (long)L * 3 * DM, w.qkv[i] + DM, w.qkv[i] + 2 * DM, 3 * DM, (long)L * 3 * DM,
                       w.img_dec[i], w.nkt_dec, w.att, DM, (long)L * DM, nullptr, nullptr, B, L, Lreg, Ar, st, rep));
      CHK(gemm_ln(Ld.out, Ld.n1, w.att, DM, w.X, DM, w.X, DM, w.tmp, rL, DM, 0, st));
      CHK(cross_and_ffn(m, sh, Ld, i, w, w.X, w.tmp, w.att, w.qc, w.ffn, rL, B, L, st));
    } else {
      // last layer: only the queried tokens of the current timestep (state tokens; Trajeglish: action tokens) of the regular slots
      CHK(launch_row_copy(w.X, DM, w.xc, DM, w.idx_state, rQ, DM, 0, st));
      CHK(launch_row_copy(w.qkv[i], 3 * DM, w.qkvc, 3 * DM, w.idx_state, rQ, 3 * DM, 0, st));
      CHK(attention_kv(amode, w.qkvc, 3 * DM, (long)Ar * 3 * DM, w.qkv[i] + DM, w.qkv[i] + 2 * DM, 3 * DM, (long)L * 3 * DM,
                       w.img_dec[i], w.nkt_dec, w.attc, DM, (long)Ar * DM, w.pos_state, nullptr, B, Ar, Lreg, Ar, st, rep));
      CHK(gemm_ln(Ld.out, Ld.n1, w.attc, DM, w.xc, DM, w.xc, DM, w.tmpc, rQ, DM, 0, st));
      CHK(cross_and_ffn(m, sh, Ld, i, w, w.xc, w.tmpc, w.attc, w.qcc, w.ffnc, rQ, B, Ar, st));
    }
  }
  // ---- predict_rtg head on the state tokens (decoder.py:74-77) / predict_action for the baselines (decoder.py:58-64)
  if (variant) return mlp_tail(m->head_action, w.xc, rQ, w.headh, logits, d.V, st);
  return mlp_tail(m->head_rtg, w.xc, rQ, w.headh, logits, d.R * d.C, st);
}
}  // namespace

extern "C" int ctrlsim_dt_forward_pass1_a(const ctrlsim_model* m, int B, int Tq, int Actx, const ctrlsim_ctx* c, void* workspace,
                                          float* rtg_logits, float* dbg_seg_emb, hipStream_t st) {
  if (!m || !c || !workspace || !rtg_logits || B < 1 || Tq < 1 || Tq > m->d.T || m->d.variant != 0) return CTRLSIM_EINVAL;
  return forward_full(m, B, Tq, Actx, c, workspace, rtg_logits, dbg_seg_emb, st);
}
extern "C" int ctrlsim_dt_forward_pass1(const ctrlsim_model* m, int B, int Tq, const ctrlsim_ctx* c, void* workspace,
                                        float* rtg_logits, float* dbg_seg_emb, hipStream_t st) {
  return m ? ctrlsim_dt_forward_pass1_a(m, B, Tq, m->d.A, c, workspace, rtg_logits, dbg_seg_emb, st) : CTRLSIM_EINVAL;
}
extern "C" int ctrlsim_dt_forward_actions(const ctrlsim_model* m, int B, int Tq, const ctrlsim_ctx* c, void* workspace,
                                          float* act_logits, hipStream_t st) {
  if (!m || !c || !workspace || !act_logits || B < 1 || Tq < 1 || Tq > m->d.T || m->d.variant == 0) return CTRLSIM_EINVAL;
  return forward_full(m, B, Tq, m->d.A, c, workspace, act_logits, nullptr, st);
}

// ------------------------------------------------------------------------------------------------ pass 2
extern "C" int ctrlsim_dt_forward_pass2_a(const ctrlsim_model* m, int B, int Tq, int Actx, int t, int N, int Tmax,
                                          const ctrlsim_ctx* c, const int* ctx_scn, const int* hist_rtg, void* workspace,
                                          float* act_logits, int cached, hipStream_t st) {
  if (!m || !c || !workspace || !act_logits || B < 1 || Tq < 1 || Tq > m->d.T || m->d.variant != 0) return CTRLSIM_EINVAL;
  const ctrlsim_dims& d = m->d;
  if (!actx_ok(d, Actx)) return CTRLSIM_EINVAL;
  const Shape sh = shape_of(d, Actx);
  // cached mode: the workspace is carved for the full window (K/V cache rows at b * rows(T) + position) and the context
  // tensors hold only the last Tn = min(Tq, 2) window rows
  const int Tw = cached ? d.T : Tq;
  const Ws w = carve(d, sh, B, Tw, static_cast<char*>(workspace));
  const int Ar = sh.Areg, L = sh.rows(Tw), rQ = B * Ar;
  const int ctx_rows = cached ? (Tq < 2 ? Tq : 2) : Tq, ti = ctx_rows - 1;
  const Rep rep{sh.rep * 3 * Tq, sh.mult, sh.lreg(Tw)};
  CHK(launch_assemble_rtg_rows(B, Ar, sh.A, ctx_rows, ti, t, N, Tmax, ctx_scn, c->slot_gid, hist_rtg, c->exist, c->tstep, m->tb,
                               m->zero_rtg, w.xc2, st));
  for (int i = 0; i < d.ND; ++i) {
    const DecLayer& Ld = m->dec[i];
    CHK(gemm(Ld.qkv, w.xc2, DM, nullptr, 0, w.qkvc, 3 * DM, rQ, 3 * DM, DM, 0, st));
    CHK(launch_row_copy(w.qkvc, 3 * DM, w.qkv[i], 3 * DM, w.idx_rtg, rQ, 3 * DM, 1, st));   // refresh the rtg rows' K/V
    CHK(kv_split_rows(w.qkvc + DM, w.qkvc + 2 * DM, 3 * DM, (long)Ar * 3 * DM, w.pos_rtg, B, Ar, w.nkt_dec, w.img_dec[i], st));
    CHK(attention_kv(1, w.qkvc, 3 * DM, (long)Ar * 3 * DM, w.qkv[i] + DM, w.qkv[i] + 2 * DM, 3 * DM, (long)L * 3 * DM,
                     w.img_dec[i], w.nkt_dec, w.attc, DM, (long)Ar * DM, w.pos_rtg, nullptr, B, Ar, Tq * Ar * 3, Ar, st, rep));   // keys: steps <= current
    CHK(gemm_ln(Ld.out, Ld.n1, w.attc, DM, w.xc2, DM, w.xc2, DM, w.tmpc, rQ, DM, 0, st));
    CHK(cross_and_ffn(m, sh, Ld, i, w, w.xc2, w.tmpc, w.attc, w.qcc, w.ffnc, rQ, B, Ar, st));
  }
  CHK(mlp_tail(m->head_action, w.xc2, rQ, w.headh, act_logits, d.V, st));
  return CTRLSIM_OK;
}
extern "C" int ctrlsim_dt_forward_pass2(const ctrlsim_model* m, int B, int Tq, int t, int N, int Tmax,
                                        const ctrlsim_ctx* c, const int* ctx_scn, const int* hist_rtg, void* workspace,
                                        float* act_logits, int cached, hipStream_t st) {
  return m ? ctrlsim_dt_forward_pass2_a(m, B, Tq, m->d.A, t, N, Tmax, c, ctx_scn, hist_rtg, workspace, act_logits, cached, st)
           : CTRLSIM_EINVAL;
}

// ------------------------------------------------------------------------------------------------ pass 1, cached
// While t < T the window starts at step 0, so the frame of a context (focal pose at window index 0), its membership
// and its map never change: the scene side is computed once (t == 0) and the decoder K/V of every layer are cached at
// fixed rows (b * rows(T) + position).  Step t only evaluates the rows whose inputs changed — the action tokens of step
// t-1 (placeholder -> applied action) and the three tokens of step t of every slot — against the cache: 4A rows instead of
// 3A*(t+1).  No other hidden state changes: an action token is visible only to later timesteps and to itself (mask closed form).
// ctx holds the window rows [max(t-1,0), t]; the workspace must be the one used at t-1 (sized with Tq = T).
extern "C" int ctrlsim_dt_forward_pass1_cached_a(const ctrlsim_model* m, int B, int t, int Actx, const ctrlsim_ctx* c,
                                                 void* workspace, float* rtg_logits, hipStream_t st) {
  if (!m || !c || !workspace || !rtg_logits || B < 1 || t < 0 || t >= m->d.T || m->d.variant != 0) return CTRLSIM_EINVAL;
  const ctrlsim_dims& d = m->d;
  if (!actx_ok(d, Actx)) return CTRLSIM_EINVAL;
  const Shape sh = shape_of(d, Actx);
  const Ws w = carve(d, sh, B, d.T, static_cast<char*>(workspace));
  const int A = sh.A, Ar = sh.Areg, P = d.P, M = P + A, Lf = sh.rows(d.T), Lreg_f = sh.lreg(d.T);
  const int Rn = (t > 0 ? 4 : 3) * A, tt_first = t > 0 ? t - 1 : 0, Tn = t + 1 - tt_first;
  const int rA = B * A, rQ = B * Ar, rN = B * Rn, rS = B * Tn * A;
  const Rep rep{sh.rep * 3 * (t + 1), sh.mult, Lreg_f};
  auto fill_cached = [&]() {
    hipLaunchKernelGGL(fill_index_cached_kernel, dim3((rN + 255) / 256), dim3(256), 0, st, B, A, Ar, Lf, Lreg_f, sh.rep_k0(d.T), t,
                       Rn, w.pos_new, w.key_new, w.src_new, w.idx_new, w.idx_state_in_new, w.pos_rtg, w.idx_rtg);
  };
  fill_cached();
  CHK(launch_in_mlp(c->st12, 12, 12, m->embed_state.l0.w, m->embed_state.l0.b, m->embed_state.ln.g, m->embed_state.ln.b,
                    w.hS, DM, rS, st));
  CHK(gemm(m->fold_state, w.hS, DM, nullptr, 0, w.S2, DM, rS, DM, DM, 0, st));
  if (t == 0) {
    CHK(launch_in_mlp(c->goal5, 5, 5, m->embed_goal.l0.w, m->embed_goal.l0.b, m->embed_goal.ln.g, m->embed_goal.ln.b, w.hG,
                      DM, rA, st));
    CHK(gemm(m->fold_goal, w.hG, DM, nullptr, 0, w.Gp, DM, rA, DM, DM, 0, st));
    const int nidx = max(max(rA, B * P), 3 * A);
    hipLaunchKernelGGL(fill_index_kernel, dim3((nidx + 255) / 256), dim3(256), 0, st, B, Ar, 3 * A, 3 * Ar, 0, 0, P, M, 0,
                       w.pos_state, w.pos_rtg, w.idx_state, w.idx_rtg, w.idx_poly, w.key_all);
    // token order of assemble_tokens at Tq = 1 is the pos_new order (regular (a, k), then the representative); it also writes
    // the initial-state rows of `src`
    CHK(launch_assemble_tokens(B, 1, A, Ar, w.S2, w.Gp, c->exist, c->act_tok, c->rtg_bin, c->tstep, m->tb, w.xn, w.src, M, P,
                               w.src_pad, st));
    CHK(scene_side(m, sh, w, c, B, nullptr, st));
    fill_cached();                                 // pos_rtg / idx_rtg for the cache layout
  } else {
    CHK(launch_assemble_rows(B, Rn, A, tt_first, Tn, w.src_new, w.S2, w.Gp, c->exist, c->act_tok, c->rtg_bin, c->tstep, m->tb,
                             w.xn, st));
  }
  for (int i = 0; i < d.ND; ++i) {
    const DecLayer& Ld = m->dec[i];
    CHK(gemm(Ld.qkv, w.xn, DM, nullptr, 0, w.qkvn, 3 * DM, rN, 3 * DM, DM, 0, st));
    CHK(launch_row_copy(w.qkvn, 3 * DM, w.qkv[i], 3 * DM, w.idx_new, rN, 3 * DM, 1, st));       // K/V (and Q) into the cache
    if (t == 0 && presplit()) {   // image tiles are read whole: stale bits beyond the written rows must at least be finite
      if (hipMemsetAsync(w.img_dec[i], 0, w.img_dec_bytes, st) != hipSuccess) return CTRLSIM_ELAUNCH;
    }
    CHK(kv_split_rows(w.qkvn + DM, w.qkvn + 2 * DM, 3 * DM, (long)Rn * 3 * DM, w.key_new, B, Rn, w.nkt_dec, w.img_dec[i], st));
    CHK(attention_kv(1, w.qkvn, 3 * DM, (long)Rn * 3 * DM, w.qkv[i] + DM, w.qkv[i] + 2 * DM, 3 * DM, (long)Lf * 3 * DM,
                     w.img_dec[i], w.nkt_dec, w.attn_n, DM, (long)Rn * DM, w.pos_new, nullptr, B, Rn, (t + 1) * Ar * 3, Ar, st, rep));
    CHK(gemm_ln(Ld.out, Ld.n1, w.attn_n, DM, w.xn, DM, w.xn, DM, w.tmpn, rN, DM, 0, st));
    CHK(cross_and_ffn(m, sh, Ld, i, w, w.xn, w.tmpn, w.attn_n, w.qcn, w.ffnn, rN, B, Rn, st));
  }
  CHK(launch_row_copy(w.xn, DM, w.xc, DM, w.idx_state_in_new, rQ, DM, 0, st));
  CHK(mlp_tail(m->head_rtg, w.xc, rQ, w.headh, rtg_logits, d.R * d.C, st));
  return CTRLSIM_OK;
}
extern "C" int ctrlsim_dt_forward_pass1_cached(const ctrlsim_model* m, int B, int t, const ctrlsim_ctx* c, void* workspace,
                                               float* rtg_logits, hipStream_t st) {
  return m ? ctrlsim_dt_forward_pass1_cached_a(m, B, t, m->d.A, c, workspace, rtg_logits, st) : CTRLSIM_EINVAL;
}
