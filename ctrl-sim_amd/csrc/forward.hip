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

#include "common.h"
#include "classes.h"
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
int launch_kv_zero_tails(int, const KvTailHost*, int, void* const*, hipStream_t);
int launch_kv_split_rows_classes(const float*, const float*, int, int, const KvRowsHost*, void*, hipStream_t);
int launch_outproj_ln_q(const float*, int, const float*, int, const void*, const float*, const float*, const float*, const void*, const float*,
                        float*, int, float*, int, int, hipStream_t);
int launch_ffn_fused_pre(const float*, int, const float*, int, const void*, const float*, const float*, const float*, const void*, const float*,
                         const void*, const float*, const float*, const float*, float*, int, int, int, hipStream_t);
int launch_ffn_fused_bf16x6(const float*, int, const void*, const float*, const void*, const float*, const float*, const float*,
                            float*, int, int, int, hipStream_t);
int launch_kv_split(const float*, const float*, int, long, int, int, int, void*, hipStream_t);
int launch_kv_split_rows(const float*, const float*, int, long, const int*, int, int, int, void*, hipStream_t);
int launch_attn_mask_tables(int, const AttnClassHost*, hipStream_t);
size_t attn_mask_table_bytes(int, int);
int launch_attention_classes(int, const float*, int, const void*, float*, int, const unsigned char*, int, const AttnClassHost*,
                             hipStream_t);
int launch_inproj_rs(const float*, int, const void*, const float*, float*, int, int, int, void*, int, int, const KvClassHost*, hipStream_t);
int launch_gemm_nt_bf16x6_kvc(const float*, int, const void*, int, int, const float*, const float*, int, float*, int, int, int, int,
                              int, const float*, const float*, void*, int, int, const KvClassHost*, hipStream_t);
int launch_in_mlp(const float*, int, int, const float*, const float*, const float*, const float*, float*, int, int,
                  hipStream_t);
int launch_row_copy(const float*, int, float*, int, const int*, int, int, int, hipStream_t);
int launch_gemm256_rows(const float*, int, const void*, int, int, const float*, float*, int, const int*, int, hipStream_t);
int launch_attention(int, const float*, int, long, const float*, const float*, int, long, float*, int, long, const int*,
                     const unsigned char*, int, int, int, int, hipStream_t);
struct EmbedTables { const float *act, *rtg_g, *rtg_v, *rtg_r, *rtg_bias, *tstep, *agent, *ln_g, *ln_b; int rtg_linear, flags; };
int launch_assemble_tokens(int, int, int, int, const float*, const float*, const float*, const int*, const int*, const int*,
                           EmbedTables, float*, float*, int, int, unsigned char*, hipStream_t);
int launch_assemble_rows(int, int, int, int, int, const int*, const float*, const float*, const float*, const int*, const int*,
                         const int*, EmbedTables, float*, hipStream_t);
int launch_assemble_rtg_rows(int, int, int, int, int, int, int, int, const int*, const int*, const int*, const float*,
                             const int*, EmbedTables, const int*, float*, hipStream_t);
struct MapPoolWeights { const float *Wc2, *Wc, *G, *ln_b, *U, *cb, *Mt, *mb; int force_pad; };
int launch_map_pool(int, int, int, int, const float*, MapPoolWeights, float*, unsigned char*, hipStream_t);
int launch_map_pool_classes(int, const int*, const int*, const long*, int, int, const float*, MapPoolWeights, float*, unsigned char*,
                            hipStream_t);
int launch_assemble_tokens_classes(int, const int*, const int*, const int*, const int*, const long*, const long*, const long*, int,
                                   const float*, const float*, const float*, const int*, const int*, const int*, EmbedTables, float*,
                                   float*, int, unsigned char*, hipStream_t);

namespace {

// ctrlsim_dims.variant 4 = the CtRL-Sim token layout and heads (variant 0) under cfg.model.attend_own_return_action (attention mask mode 5)
inline int tok_variant(int v) { return v == 4 ? 0 : v; }

struct Lin {
  const float* w; const float* b;
  const void* w3s[2] = {nullptr, nullptr};      // operand planes per split scheme (OPT_SPLIT 0 / 1)
  int ntot = 0; int n0 = 0;
  const void* wblk = nullptr;                   // 32-column operand blocks from column n0 on (two-fp16-plane scheme; pack.py:row_blocks)
  const void* w3() const { return w3s[ctrlsim_option(OPT_SPLIT) ? 1 : 0]; }
};
struct LNp { const float* g; const float* b; };
struct Mlp { Lin l0; LNp ln; Lin l3; };
struct FfnPlanes { const void *w1[2] = {nullptr, nullptr}, *w2[2] = {nullptr, nullptr}, *wop = nullptr, *w1q = nullptr; };   // wop / w1q: pack.py:ffn_planes_pre
struct EncLayer { Lin qkv, out, lin1, lin2; LNp n1, n2; FfnPlanes fp; };
// sq / skv: the query rows / the key + value rows of the self-attention in_proj as Linears of their own (the last decoder layer of a full
// pass projects keys and values of every token but queries of the A queried tokens only)
struct DecLayer {
  Lin qkv, sq, skv, out, cq, ckv, cout, lin1, lin2; LNp n1, n2, n3; FfnPlanes fp;
  const void *sq_wop = nullptr, *sq_wqp = nullptr;      // pack.py:outproj_q_planes(self_attn.out_proj, cross-attention Wq)
};

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
  Mlp head_action, head_rtg, head_fut;
  bool has_fut = false;
  hipEvent_t ev_tail = nullptr;           // orders the few-row tail of a first pass behind its full-row part when the tail runs on another stream
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
  if (dims->D != DM || dims->H != NHEAD || dims->A < 1 || dims->A > 64 || dims->variant < 0 || dims->variant > 4)
    return CTRLSIM_EINVAL;
  // flags (include/ctrlsim.h): 1 no_actions, 2 use_map = False, 4 encode_initial_state = False — CtRL-Sim token layout only; without a map AND
  // without the initial states the reference has no scene memory to build (modules/encoder.py:155-170 reads an undefined tensor)
  if ((dims->flags & ~7) || (dims->flags && tok_variant(dims->variant) != 0) || (dims->flags & 6) == 6) return CTRLSIM_EINVAL;
  std::unordered_map<std::string, const float*> tab;
  for (int i = 0; i < n; ++i) tab[names[i]] = dev_weights + offsets[i];
  bool ok = true;
  auto P = [&](const std::string& k) -> const float* {
    auto it = tab.find(k);
    if (it == tab.end()) { ok = false; return nullptr; }
    return it->second;
  };
  auto PX = [&](const std::string& k) -> const void* {     // optional extra operand images
    auto it = tab.find(k);
    return it == tab.end() ? nullptr : static_cast<const void*>(it->second);
  };
  auto mk = [&](const float* w, const float* b, const std::string& planes_of, int ntot, int n0) {   // operand planes of both schemes
    Lin L{w, b};
    L.w3s[0] = PX(planes_of + "#pl0"); L.w3s[1] = PX(planes_of + "#pl1");
    L.ntot = ntot; L.n0 = n0;
    if (const void* blk = PX(planes_of + "#blk#pl1"))       // [cb][2 planes][16][2][32][8] 16-bit words: 32 KB per 32 columns
      L.wblk = static_cast<const char*>(blk) + (size_t)(n0 / 32) * 32768;
    return L;
  };
  auto ffnp = [&](const std::string& p) {
    FfnPlanes f;
    f.w1[0] = PX(p + ".ffn#w1p#pl0"); f.w1[1] = PX(p + ".ffn#w1p#pl1");
    f.w2[0] = PX(p + ".ffn#w2p#pl0"); f.w2[1] = PX(p + ".ffn#w2p#pl1");
    f.wop = PX(p + ".ffn#wop#pl1"); f.w1q = PX(p + ".ffn#w1q#pl1");
    return f;
  };
  auto lin = [&](const std::string& k) { return mk(P(k + ".weight"), P(k + ".bias"), k + ".weight", 0, 0); };
  auto lnp = [&](const std::string& k) { return LNp{P(k + ".weight"), P(k + ".bias")}; };
  auto mlp = [&](const std::string& k) { return Mlp{lin(k + ".mlp.0"), lnp(k + ".mlp.1"), lin(k + ".mlp.3")}; };
  ctrlsim_model* m = new ctrlsim_model();
  m->d = *dims;
  m->embed_state = mlp("encoder.embed_state");
  m->embed_goal = mlp("encoder.embed_goal");
  m->fold_state = mk(P("fold.embed_state.w"), nullptr, "fold.embed_state.w", 0, 0);
  m->fold_goal = mk(P("fold.embed_goal.w"), P("fold.embed_goal.b"), "fold.embed_goal.w", 0, 0);
  m->tb = EmbedTables{P("encoder.embed_action.weight"), P("fold.rtg_table_goal"), P("fold.rtg_table_veh"),
                      P("fold.rtg_table_road"), P("fold.rtg_bias"), P("encoder.embed_timestep.weight"),
                      P("encoder.embed_agent_id.weight"), P("encoder.embed_ln.weight"), P("encoder.embed_ln.bias"),
                      dims->variant == 3 ? 1 : 0, dims->flags & 5};
  const std::string me = "encoder.map_encoder.";
  m->mp = MapPoolWeights{P("fold.map.Wc2"), P("fold.map.Wc"), P("fold.map.G"), P(me + "road_pts_encoder.mlp.1.bias"),
                         P("fold.map.U"), P("fold.map.cb"), P("fold.map.Mt"), P("fold.map.mb"), (dims->flags & 2) ? 1 : 0};
  m->map_out = lin(me + "road_pts_attn_layer.out_proj");
  m->map_n1 = lnp(me + "norm1");
  m->map_n2 = lnp(me + "norm2");
  m->map_feats = mlp(me + "map_feats");
  m->road_type = mlp(me + "road_type_encoder");
  m->road_fuse = mlp(me + "road_road_type_encoder");
  for (int i = 0; i < dims->NE; ++i) {
    const std::string p = "encoder.transformer_encoder.layers." + std::to_string(i);
    EncLayer L;
    L.qkv = mk(P(p + ".self_attn.in_proj_weight"), P(p + ".self_attn.in_proj_bias"), p + ".self_attn.in_proj_weight", 3 * DM, 0);
    L.out = lin(p + ".self_attn.out_proj");
    L.lin1 = lin(p + ".linear1");
    L.lin2 = lin(p + ".linear2");
    L.n1 = lnp(p + ".norm1");
    L.n2 = lnp(p + ".norm2");
    L.fp = ffnp(p);
    m->enc.push_back(L);
  }
  for (int i = 0; i < dims->ND; ++i) {
    const std::string p = "decoder.transformer_decoder.layers." + std::to_string(i);
    DecLayer L;
    L.qkv = mk(P(p + ".self_attn.in_proj_weight"), P(p + ".self_attn.in_proj_bias"), p + ".self_attn.in_proj_weight", 3 * DM, 0);
    {
      const float* sw = P(p + ".self_attn.in_proj_weight");
      const float* sb = P(p + ".self_attn.in_proj_bias");
      L.sq = mk(sw, sb, p + ".self_attn.in_proj_weight", 3 * DM, 0);
      L.skv = mk(sw ? sw + DM * DM : nullptr, sb ? sb + DM : nullptr, p + ".self_attn.in_proj_weight", 3 * DM, DM);
    }
    L.out = lin(p + ".self_attn.out_proj");
    const float* cw = P(p + ".multihead_attn.in_proj_weight");
    const float* cb = P(p + ".multihead_attn.in_proj_bias");
    L.cq = mk(cw, cb, p + ".multihead_attn.in_proj_weight", 3 * DM, 0);
    L.ckv = mk(cw ? cw + DM * DM : nullptr, cb ? cb + DM : nullptr, p + ".multihead_attn.in_proj_weight", 3 * DM, DM);
    L.cout = lin(p + ".multihead_attn.out_proj");
    L.lin1 = lin(p + ".linear1");
    L.lin2 = lin(p + ".linear2");
    L.n1 = lnp(p + ".norm1");
    L.n2 = lnp(p + ".norm2");
    L.n3 = lnp(p + ".norm3");
    L.fp = ffnp(p);
    L.sq_wop = PX(p + ".selfq#wop#pl1"); L.sq_wqp = PX(p + ".selfq#wqp#pl1");
    m->dec.push_back(L);
  }
  m->head_action = mlp("decoder.predict_action");
  if (tok_variant(dims->variant) == 0) m->head_rtg = mlp("decoder.predict_rtg");      // the IL / Trajeglish models have no such head
  if (tab.count("decoder.predict_future_states.mlp.0.weight")) {        // model.predict_future_states (decoder.py:29-30): training-time
    m->head_fut = mlp("decoder.predict_future_states");                 // auxiliary head, only read by ctrlsim_forward_all
    m->has_fut = true;
  }
  m->zero_rtg[0] = 0; m->zero_rtg[1] = 35; m->zero_rtg[2] = 35;
  if (!ok) { delete m; return CTRLSIM_EINVAL; }
  if (hipEventCreateWithFlags(&m->ev_tail, hipEventDisableTiming) != hipSuccess) { delete m; return CTRLSIM_ELAUNCH; }
  *out = m;
  return CTRLSIM_OK;
}

extern "C" void ctrlsim_model_destroy(ctrlsim_model* m) {
  if (m && m->ev_tail) (void)hipEventDestroy(m->ev_tail);
  delete m;
}

// ------------------------------------------------------------------------------------------------ batch of context classes
namespace {
// Shape of a CLASS of contexts.  Token rows of agent slots that do not exist anywhere in the window are all equal (every
// embedding is multiplied by the existence flag before embed_ln, modules/encoder.py:127-133) and stay equal through the decoder
// (no key padding on the targets; the structured mask treats all of them alike), so a context with n < A vehicles can be
// evaluated with Actx >= n + 1 slots: Areg = Actx - 1 regular slots, and ONE representative slot standing for the
// mult = A - Areg padded slots of the reference's 24-slot layout (attention_bf16x6.hip: multiplicity of its keys).  A context's
// L token rows: regular (tt, a < Areg, k) at (tt*Areg + a)*3 + k, representative (tt, k) at Lreg + 3*tt + k; its keys sit in
// the tiles from key rep_k0 = 64*ceil(Lreg/64) on.  Actx == A: the plain layout (rep = 0).  Exact in real arithmetic.
// A model BATCH is up to MAXC classes (the engine sorts the contexts of a step by vehicle count); every row-wise kernel
// (Linear, LayerNorm, feed-forward) runs ONCE over the rows of all classes, the two shape-aware kernels (attention, the
// K/V-image epilogue of the QKV projection) take a class table, the small index / gather kernels run per class.
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
struct Cls {
  Shape sh;
  int B, L, Lreg, M, nkt_dec, nkt_mem;
  // first row of this class in the buffers indexed by: token rows, (b, tt, slot), (b, slot), scene rows, polylines, query rows, new rows
  long rL, rS, rA, rM, rP, rQ, rN;
  long tile_dec, tile_mem;                 // first image tile (per layer buffer)
  int ioff;                                // offset of the class's [A]-sized index lists; 4 * ioff for the [4A]-sized ones
  long koff;                               // offset of its key_all list
  const ctrlsim_ctx* ctx;
};
struct Batch {
  bool contig = false;                    // the classes' context tensors lie back to back (ctx_contiguous): merged launches
  int n, Btot;
  Cls c[MAXC];
  long rL, rS, rA, rM, rP, rQ, rN, tiles_dec, tiles_mem, isum, ksum;
};
// Tw: window steps of the row / image layout (T for the K/V-cached phase); Tn: window rows held by the context tensors
int make_batch(const ctrlsim_dims& d, int n, const int* B, const int* A, const ctrlsim_ctx* ctx, int Tw, int Tn, int Rn_mul,
               Batch& bt) {
  if (n < 1 || n > MAXC || !B || !A || !ctx) return CTRLSIM_EINVAL;
  bt = Batch{};
  for (int k = 0; k < n; ++k) {
    if (B[k] <= 0) continue;
    if (A[k] < 2 || A[k] > d.A) return CTRLSIM_EINVAL;
    Cls& c = bt.c[bt.n++];
    c.sh = shape_of(d, A[k]);
    c.B = B[k]; c.L = c.sh.rows(Tw); c.Lreg = c.sh.lreg(Tw);
    c.M = (d.P + A[k] + 3) & ~3;             // scene rows per context: polylines, initial states, key-padded filler up to a multiple of 4
                                              // (the K/V-image epilogue of the memory projections works on key quads)
    c.nkt_dec = c.sh.nkt(Tw); c.nkt_mem = (c.M + 63) / 64;
    c.rL = bt.rL; c.rS = bt.rS; c.rA = bt.rA; c.rM = bt.rM; c.rP = bt.rP; c.rQ = bt.rQ; c.rN = bt.rN;
    c.tile_dec = bt.tiles_dec; c.tile_mem = bt.tiles_mem; c.ioff = (int)bt.isum; c.koff = bt.ksum;
    c.ctx = ctx + k;
    bt.rL += (long)c.B * c.L; bt.rS += (long)c.B * Tn * A[k]; bt.rA += (long)c.B * A[k]; bt.rM += (long)c.B * c.M;
    bt.rP += (long)c.B * d.P; bt.rQ += (long)c.B * c.sh.Areg; bt.rN += (long)c.B * Rn_mul * A[k];
    bt.tiles_dec += (long)c.B * NHEAD * c.nkt_dec; bt.tiles_mem += (long)c.B * NHEAD * c.nkt_mem;
    bt.isum += A[k]; bt.ksum += c.L;
    bt.Btot += c.B;
  }
  return bt.n ? CTRLSIM_OK : CTRLSIM_EINVAL;
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
  // layer, scene encoder (reused by its layers)
  void *img_dec[8], *img_mem[8], *img_enc;
  size_t img_dec_bytes;
  // visibility-mask tables of the classes (attention_bf16x6.hip), rebuilt by every full forward pass: class k at mask_tbl[k]
  char* mask_tbl[MAXC];
  size_t bytes;
};

// Sizes by the batch totals (every buffer holds the classes one after the other); Tn = window rows of the context tensors
Ws carve(const ctrlsim_dims& d, const Batch& bt, char* base) {
  Ws w;
  size_t off = 0;
  auto take = [&](size_t nbytes) -> char* {
    char* p = base ? base + off : nullptr;
    off += (nbytes + 255) & ~size_t(255);
    return p;
  };
  const size_t rL = bt.rL, rM = bt.rM, rA = bt.rA, rS = bt.rS, rP = bt.rP;
  auto F = [&](size_t rows, size_t cols) { return reinterpret_cast<float*>(take(rows * cols * sizeof(float))); };
  auto I = [&](size_t n) { return reinterpret_cast<int*>(take(n * sizeof(int))); };
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
  w.pos_state = I(bt.isum); w.pos_rtg = I(bt.isum);
  w.idx_state = I(rA); w.idx_rtg = I(rA); w.idx_poly = I(rP); w.key_all = I(bt.ksum);
  const size_t rN = rA * 4;
  w.xn = F(rN, DM); w.tmpn = F(rN, DM); w.attn_n = F(rN, DM); w.qkvn = F(rN, 3 * DM); w.qcn = F(rN, DM); w.ffnn = F(rN, d.F);
  w.pos_new = I(4 * bt.isum); w.key_new = I(4 * bt.isum); w.src_new = I(4 * bt.isum);
  w.idx_new = I(rN); w.idx_state_in_new = I(rA);
  const size_t tile_bytes = split_kimg() * 2;   // 8 KB per plane pair of a (context, head, tile) image
  w.img_dec_bytes = (size_t)bt.tiles_dec * tile_bytes;
  for (int i = 0; i < d.ND; ++i) w.img_dec[i] = take(w.img_dec_bytes);
  for (int i = 0; i < d.ND; ++i) w.img_mem[i] = take((size_t)bt.tiles_mem * tile_bytes);
  w.img_enc = take((size_t)bt.tiles_mem * tile_bytes);
  for (int k = 0; k < MAXC; ++k) w.mask_tbl[k] = k < bt.n ? take(attn_mask_table_bytes(bt.c[k].L, bt.c[k].nkt_dec)) : nullptr;
  w.bytes = off;
  return w;
}

// qoff: token type whose rows feed the first pass's head (0 = state tokens; 2 = action tokens, Trajeglish).  Index lists of ONE
// class: the Areg current-step query rows per context (positions in the context's row order — regular rows: position == key —
// and global row indices), the polyline rows of the scene-encoder source, key_all: row -> key position in the K/V images.
__global__ void fill_index_kernel(int B, int Areg, int L, int Lreg, int rep_k0, int ti, int P, int M, int qoff, long rL, long rM,
                                  int* pos_state, int* pos_rtg, int* idx_state, int* idx_rtg, int* idx_poly, int* key_all) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * P) idx_poly[i] = (int)rM + (i / P) * M + (i % P);
  if (i < Areg) { pos_state[i] = (ti * Areg + i) * 3 + qoff; pos_rtg[i] = (ti * Areg + i) * 3 + 1; }
  if (i < B * Areg) {
    const int b = i / Areg, a = i - b * Areg;
    idx_state[i] = (int)rL + b * L + (ti * Areg + a) * 3 + qoff;
    idx_rtg[i] = (int)rL + b * L + (ti * Areg + a) * 3 + 1;
  }
  if (i < L) key_all[i] = i < Lreg ? i : rep_k0 + (i - Lreg);
}

// y = act(x W^T + b [+ R]) through the bf16x6 MFMA kernel when the packed planes exist, else the f32-input MFMA kernel
int gemm(const Lin& L, const float* x, int ldx, const float* R, int ldr, float* y, int ldy, int rows, int n, int k, int relu,
         hipStream_t st) {
  if (L.w3() && k % 32 == 0 && ctrlsim_option(OPT_GEMM_IMPL) == 1) {
    // tall plain Linears whose weight travels as 32-column blocks (the cross-attention query projection): row-stationary kernel
    if (L.wblk && !R && !relu && k == DM && n == DM && rows >= 16384 && !(L.n0 & 31) && ctrlsim_option(OPT_SPLIT) &&
        (ctrlsim_option(OPT_GEMM_WS) & 16))
      return launch_inproj_rs(x, ldx, L.wblk, L.b, y, ldy, rows, n, nullptr, n, 0, nullptr, st);
    return launch_gemm_nt_bf16x6(x, ldx, L.w3(), L.ntot ? L.ntot : n, L.n0, L.b, R, ldr, y, ldy, rows, n, k, relu, nullptr,
                                 nullptr, st);
  }
  return launch_gemm_nt(x, ldx, L.w, k, L.b, R, ldr, y, ldy, rows, n, k, relu, st);
}

// y = [relu] LayerNorm(x W^T + b [+ R]) for the 256-wide blocks: one kernel on the bf16x6 path (LN in the GEMM epilogue;
// y may alias R), GEMM into `tmp` + layernorm256 on the f32-input path
int gemm_ln(const Lin& L, const LNp& n, const float* x, int ldx, const float* R, int ldr, float* y, int ldy, float* tmp,
            int rows, int k, int relu, hipStream_t st) {
  if (L.w3() && k % 32 == 0 && ctrlsim_option(OPT_GEMM_IMPL) == 1)
    return launch_gemm_nt_bf16x6(x, ldx, L.w3(), L.ntot ? L.ntot : DM, L.n0, L.b, R, ldr, y, ldy, rows, DM, k, relu, n.g, n.b, st);
  CHK(launch_gemm_nt(x, ldx, L.w, k, L.b, R, ldr, tmp, DM, rows, DM, k, 0, st));
  return launch_layernorm256(tmp, DM, nullptr, 0, n.g, n.b, y, ldy, rows, relu, st);
}

// x <- LayerNorm(x + linear2(relu(linear1(x)))): one fused kernel (hidden tile in registers) or Linear + Linear/LN
int ffn_block(const Lin& l1, const Lin& l2, const LNp& n, const FfnPlanes& fp, float* x, float* hidden, float* tmp,
              int rows, int F, hipStream_t st) {
  const int sch = ctrlsim_option(OPT_SPLIT) ? 1 : 0;
  const void *w1p = fp.w1[sch], *w2p = fp.w2[sch];
  if (w1p && w2p && !(F & 31) && ctrlsim_option(OPT_FFN_FUSED) >= 1 && ctrlsim_option(OPT_GEMM_IMPL) == 1)
    return launch_ffn_fused_bf16x6(x, DM, w1p, l1.b, w2p, l2.b, n.g, n.b, x, DM, rows, F, st);
  CHK(gemm(l1, x, DM, nullptr, 0, hidden, F, rows, F, DM, 1, st));
  return gemm_ln(l2, n, hidden, F, x, DM, x, DM, tmp, rows, F, 0, st);
}

// x <- FFN block( LayerNorm_o(x + att Wo^T + bo) ): the attention out-projection + residual + LayerNorm and the feed-forward block behind it —
// ONE kernel where the two-plane scheme and the images of pack.py:ffn_planes_pre are there (option 3 = 2), the two kernels otherwise
int outproj_ln_ffn(const Lin& lo, const LNp& no, const Lin& l1, const Lin& l2, const LNp& n, const FfnPlanes& fp, const float* att, float* x,
                   float* hidden, float* tmp, int rows, int F, hipStream_t st) {
  if (fp.wop && fp.w1q && fp.w2[1] && !(F & 31) && F <= 3072 && ctrlsim_option(OPT_SPLIT) && ctrlsim_option(OPT_FFN_FUSED) >= 2 &&
      ctrlsim_option(OPT_GEMM_IMPL) == 1)
    return launch_ffn_fused_pre(att, DM, x, DM, fp.wop, lo.b, no.g, no.b, fp.w1q, l1.b, fp.w2[1], l2.b, n.g, n.b, x, DM, rows, F, st);
  CHK(gemm_ln(lo, no, att, DM, x, DM, x, DM, tmp, rows, DM, 0, st));
  return ffn_block(l1, l2, n, fp, x, hidden, tmp, rows, F, st);
}

inline bool presplit() { return ctrlsim_option(OPT_ATTN_IMPL) == 1; }

// What the queries / keys of an attention call are, per class
enum QKind { Q_ALL, Q_SCENE, Q_STATE, Q_RTG, Q_NEW };   // all L token rows | the M scene rows | the Areg state (first-pass) rows |
                                                         // the Areg rtg rows | the cached path's new rows
struct AttnCall {
  int mode;                      // 0 key padding (memory / scene keys), >= 1 causal
  QKind q;
  const float* Q; int ldq;       // Q rows in class order (Q_ALL: the token rows; else the compact rows)
  const float* Kf; const float* Vf; int ldkv;   // fp32 K / V rows (f32-input fallback only)
  const void* img; bool mem;     // images: decoder self-attention tiles (mem = false) or memory / scene tiles
  float* O;
  int Tk;                        // causal: window steps whose keys are attended (Lk = Tk * 3 * Areg, rep_keys = 3 * Tk)
  int Tw;                        // window steps of the row layout (rep_pos0 = lreg(Tw))
  int Rn_mul;                    // Q_NEW: rows per context = Rn_mul * A
  bool tbl = false;              // Q_ALL, causal: the classes' mask tables (Ws::mask_tbl) were built for this pass (build_mask_tables)
};
int attention(const ctrlsim_dims& d, const Batch& bt, const Ws& w, const AttnCall& a, hipStream_t st) {
  AttnClassHost h[MAXC];
  for (int k = 0; k < bt.n; ++k) {
    const Cls& c = bt.c[k];
    const Shape& sh = c.sh;
    AttnClassHost& x = h[k];
    x = AttnClassHost{};
    x.B = c.B;
    x.A = sh.Areg;
    long qrow0 = 0;
    switch (a.q) {
      case Q_ALL: x.Lq = c.L; qrow0 = c.rL; x.q_pos = nullptr; break;
      case Q_SCENE: x.Lq = c.M; qrow0 = c.rM; x.q_pos = nullptr; break;
      case Q_STATE: x.Lq = sh.Areg; qrow0 = c.rQ; x.q_pos = w.pos_state + c.ioff; break;
      case Q_RTG: x.Lq = sh.Areg; qrow0 = c.rQ; x.q_pos = w.pos_rtg + c.ioff; break;
      case Q_NEW: x.Lq = a.Rn_mul * sh.A; qrow0 = c.rN; x.q_pos = w.pos_new + 4 * c.ioff; break;
    }
    x.q_row0 = qrow0; x.o_row0 = qrow0;
    x.q_bs = (long)x.Lq * a.ldq; x.o_bs = (long)x.Lq * DM;
    if (a.mode == 0) {
      x.Lk = c.M; x.nkt = c.nkt_mem; x.img_tile0 = c.tile_mem; x.pad_off = c.rM; x.q_pos = nullptr;
    } else {
      x.Lk = a.Tk * 3 * sh.Areg; x.rep_keys = sh.rep * 3 * a.Tk; x.rep_mult = sh.mult; x.rep_pos0 = sh.lreg(a.Tw);
      x.nkt = c.nkt_dec; x.img_tile0 = c.tile_dec; x.pad_off = 0;
      if (a.tbl && a.q == Q_ALL) x.mask_tbl = w.mask_tbl[k];
    }
  }
  if (presplit()) return launch_attention_classes(a.mode, a.Q, a.ldq, a.img, a.O, DM, w.src_pad, bt.n, h, st);
  // f32-input MFMA path: one launch per class, plain layout only
  for (int k = 0; k < bt.n; ++k) {
    const Cls& c = bt.c[k];
    const AttnClassHost& x = h[k];
    if (x.rep_keys) return CTRLSIM_EINVAL;
    const long krow0 = a.mode == 0 ? c.rM : c.rL, kbs = (long)(a.mode == 0 ? c.M : c.L) * a.ldkv;
    CHK(launch_attention(a.mode, a.Q + x.q_row0 * a.ldq, a.ldq, x.q_bs, a.Kf + krow0 * a.ldkv, a.Vf + krow0 * a.ldkv, a.ldkv, kbs,
                         a.O + x.o_row0 * DM, DM, x.o_bs, x.q_pos, a.mode == 0 ? w.src_pad + c.rM : nullptr, c.B, x.Lq, x.Lk, x.A,
                         st));
  }
  return 0;
}

// The classes' visibility-mask tables for the causal launches over all token rows of a full pass over Tq window steps (one small launch)
int build_mask_tables(const Batch& bt, const Ws& w, int Tq, hipStream_t st) {
  AttnClassHost h[MAXC];
  for (int k = 0; k < bt.n; ++k) {
    const Cls& c = bt.c[k];
    h[k] = AttnClassHost{};
    h[k].B = c.B; h[k].Lq = c.L; h[k].Lk = Tq * 3 * c.sh.Areg; h[k].A = c.sh.Areg; h[k].rep_keys = c.sh.rep * 3 * Tq;
    h[k].rep_mult = c.sh.mult; h[k].rep_pos0 = c.sh.lreg(Tq); h[k].nkt = c.nkt_dec; h[k].mask_tbl = w.mask_tbl[k];
  }
  return launch_attn_mask_tables(bt.n, h, st);
}

// Linear whose last 512 output columns are attention keys / values: y[:, :kcol0] as fp32 rows, K / V as split images straight from
// the GEMM epilogue when both bf16x6 kernels are selected (the fp32 K / V columns of y are then NOT written); otherwise GEMM +
// a split pass per class.  mem: the scene / memory rows (M per context, plain key order) instead of the token rows.
// The tails the row-wise epilogues leave in the last tiles of the key regions: classes' entries for the token rows (mem = false) or the
// scene rows (mem = true), applied to nimg image sets of that geometry in ONE launch (tails of whole sub-tiles are skipped: nobody reads them)
int zero_kv_tails(const Batch& bt, bool mem, int nimg, void* const* imgs, hipStream_t st) {
  KvTailHost tails[2 * MAXC];
  int nt = 0;
  for (int k = 0; k < bt.n; ++k) {
    const Cls& c = bt.c[k];
    const int Lk = mem ? c.M : c.L, Lreg = mem ? c.M : c.Lreg, nkt = mem ? c.nkt_mem : c.nkt_dec;
    const long tile0 = mem ? c.tile_mem : c.tile_dec;
    tails[nt++] = KvTailHost{c.B, 0, Lreg, nkt, tile0};
    if (Lreg < Lk) tails[nt++] = KvTailHost{c.B, c.sh.rep_k0(c.Lreg / (3 * c.sh.Areg)), Lk - Lreg, nkt, tile0};
  }
  return launch_kv_zero_tails(nt, tails, nimg, imgs, st);
}

int gemm_kv(const ctrlsim_dims& d, const Batch& bt, const Ws& w, const Lin& L, const float* x, float* y, int ldy, int n, int kcol0,
            void* img, bool mem, hipStream_t st, bool tails_done = false) {
  const long rows = mem ? bt.rM : bt.rL;
  bool fused = presplit() && L.w3() && ctrlsim_option(OPT_GEMM_IMPL) == 1;
  KvClassHost kc[MAXC];
  for (int k = 0; k < bt.n; ++k) {
    const Cls& c = bt.c[k];
    const int Lk = mem ? c.M : c.L, Lreg = mem ? c.M : c.Lreg;
    kc[k] = KvClassHost{c.B, Lk, Lreg, mem ? 0 : c.sh.rep_k0(c.Lreg / (3 * c.sh.Areg)), mem ? c.nkt_mem : c.nkt_dec,
                        mem ? c.tile_mem : c.tile_dec};
    fused = fused && !(Lk & 3) && !(Lreg & 3) && Lk >= 32;
  }
  const size_t KIMG = split_kimg();      // 16-bit elements per tile
  if (fused) {
    // the epilogue writes rows: the tails of the last tiles of both key regions stay (zeroed here, unless the caller did it for
    // several image sets at once)
    if (!tails_done) { void* one[1] = {img}; CHK(zero_kv_tails(bt, mem, 1, one, st)); }
    if (L.wblk && ctrlsim_option(OPT_SPLIT) && (ctrlsim_option(OPT_GEMM_WS) & 8) && !(L.n0 & 31) && !(n & 31) && n <= 3 * DM && !(kcol0 & 31))
      return launch_inproj_rs(x, DM, L.wblk, L.b, y, ldy, (int)rows, n, img, kcol0, bt.n, kc, st);
    return launch_gemm_nt_bf16x6_kvc(x, DM, L.w3(), L.ntot ? L.ntot : n, L.n0, L.b, nullptr, 0, y, ldy, (int)rows, n, DM, 0, nullptr,
                                     nullptr, img, kcol0, bt.n, kc, st);
  }
  CHK(gemm(L, x, DM, nullptr, 0, y, ldy, (int)rows, n, DM, 0, st));
  if (!presplit()) return 0;
  for (int k = 0; k < bt.n; ++k) {
    const Cls& c = bt.c[k];
    const long r0 = mem ? c.rM : c.rL;
    unsigned short* base = static_cast<unsigned short*>(img) + (size_t)kc[k].tile0 * KIMG;
    const float* Kp = y + r0 * ldy + kcol0;
    if (kc[k].Lreg == kc[k].L) {
      CHK(launch_kv_split(Kp, Kp + DM, ldy, (long)kc[k].L * ldy, c.B, kc[k].L, kc[k].nkt, base, st));
    } else {
      if (hipMemsetAsync(base, 0, (size_t)c.B * NHEAD * kc[k].nkt * KIMG * sizeof(unsigned short), st) != hipSuccess) return CTRLSIM_ELAUNCH;
      CHK(launch_kv_split_rows(Kp, Kp + DM, ldy, (long)kc[k].L * ldy, w.key_all + c.koff, c.B, kc[k].L, kc[k].nkt, base, st));
    }
  }
  return 0;
}

// index lists of the cached path (one class): the new rows of step t are the action tokens of step t-1 (t > 0) and the three tokens
// of step t, of the Areg regular slots and then of the representative; cache rows live at rL + b * Lf + position with Lf = rows
// of the full window.  pos_new = position in the row order (= the query position the mask uses), key_new = key in the K/V
// images, src_new = ((tt - tt_first) * Actx + slot) * 3 + k in the context tensors.
__global__ void fill_index_cached_kernel(int B, int Actx, int Areg, int Lf, int Lreg_f, int rep_k0, int t, int Rn, long rL, long rN,
                                         int* pos_new, int* key_new, int* src_new, int* idx_new, int* idx_state_in_new,
                                         int* pos_rtg, int* idx_rtg) {
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
    idx_new[i] = (int)rL + (i / Rn) * Lf + pos_of(tt, a, k);
  }
  if (i < B * Areg) {
    const int b = i / Areg, a = i - b * Areg;
    idx_state_in_new[i] = (int)rN + b * Rn + off + 3 * a;
    idx_rtg[i] = (int)rL + b * Lf + (t * Areg + a) * 3 + 1;
  }
}

int mlp_tail(const Mlp& m, const float* h_in, int rows, float* hid, float* out, int n_out, hipStream_t st, int ld_in = DM) {
  // Linear(256->256) -> LN -> ReLU -> Linear(256->n_out)
  CHK(gemm_ln(m.l0, m.ln, h_in, ld_in, nullptr, 0, hid, DM, hid, rows, DM, 1, st));
  CHK(gemm(m.l3, hid, DM, nullptr, 0, out, n_out, rows, n_out, DM, 0, st));
  return 0;
}

// what follows a decoder layer's self-attention, shared by the full-row and compact-row paths (post-LN): out-projection + residual + norm1
// and the cross-attention query projection (ONE kernel with option 3 = 3 and the images of pack.py:outproj_q_planes), cross-attention,
// its out-projection + norm2 + feed-forward block.  att holds the self-attention output on entry.
int cross_and_ffn(const ctrlsim_model* m, const Batch& bt, const DecLayer& Ld, int layer, const Ws& w, float* x, float* tmp,
                  float* att, float* qc, float* ffn, long rows, QKind q, int Rn_mul, hipStream_t st) {
  const ctrlsim_dims& d = m->d;
  if (Ld.sq_wop && Ld.sq_wqp && ctrlsim_option(OPT_SPLIT) && ctrlsim_option(OPT_FFN_FUSED) >= 3 && ctrlsim_option(OPT_GEMM_IMPL) == 1) {
    CHK(launch_outproj_ln_q(att, DM, x, DM, Ld.sq_wop, Ld.out.b, Ld.n1.g, Ld.n1.b, Ld.sq_wqp, Ld.cq.b, x, DM, qc, DM, (int)rows, st));
  } else {
    CHK(gemm_ln(Ld.out, Ld.n1, att, DM, x, DM, x, DM, tmp, (int)rows, DM, 0, st));
    CHK(gemm(Ld.cq, x, DM, nullptr, 0, qc, DM, (int)rows, DM, DM, 0, st));
  }
  CHK(attention(d, bt, w, AttnCall{0, q, qc, DM, w.memkv[layer], w.memkv[layer] + DM, 2 * DM, w.img_mem[layer], true, att, 0, 0,
                                   Rn_mul}, st));
  CHK(outproj_ln_ffn(Ld.cout, Ld.n2, Ld.lin1, Ld.lin2, Ld.n3, Ld.fp, att, x, ffn, tmp, (int)rows, d.F, st));
  return 0;
}
// The engine carves the context tensors of a batch's classes out of shared buffers back to back (CtxBuffers.class_structs);
// then the per-(context, step, slot) / per-polyline kernels can run ONCE over all classes instead of once per class — a class of
// a model batch is too small a grid to fill the chip.  Tn = window rows held by the context tensors.
bool ctx_contiguous(const ctrlsim_dims& d, const Batch& bt, int Tn) {
  for (int k = 0; k + 1 < bt.n; ++k) {
    const Cls& c = bt.c[k];
    const ctrlsim_ctx* a = c.ctx; const ctrlsim_ctx* b = bt.c[k + 1].ctx;
    const long rows = (long)c.B * Tn * c.sh.A;
    if (b->st12 != a->st12 + rows * 12 || b->exist != a->exist + rows || b->act_tok != a->act_tok + rows ||
        b->rtg_bin != a->rtg_bin + rows * 3 || b->tstep != a->tstep + (long)c.B * Tn || b->goal5 != a->goal5 + (long)c.B * c.sh.A * 5 ||
        b->road_pts != a->road_pts + (long)c.B * d.P * d.NP * 3 || b->road_types != a->road_types + (long)c.B * d.P * 8)
      return false;
  }
  return true;
}

// map encoder + scene encoder + per-layer memory K/V (everything that only depends on the frame of the context)
int scene_side(const ctrlsim_model* m, const Batch& bt, const Ws& w, float* dbg_seg_emb, hipStream_t st) {
  const ctrlsim_dims& d = m->d;
  const int P = d.P, rM = (int)bt.rM, rP = (int)bt.rP;
  // ---- map encoder (map_encoder.py:34-53): rows of `src` 0..P-1 per context
  if (bt.contig) {
    int Bk[MAXC], Mk[MAXC];
    long pad0[MAXC];
    for (int k = 0; k < bt.n; ++k) { Bk[k] = bt.c[k].B; Mk[k] = bt.c[k].M; pad0[k] = bt.c[k].rM; }
    CHK(launch_map_pool_classes(bt.n, Bk, Mk, pad0, P, d.NP, bt.c[0].ctx->road_pts, m->mp, w.attn_pre, w.src_pad, st));
    CHK(launch_in_mlp(bt.c[0].ctx->road_types, 8, 8, m->road_type.l0.w, m->road_type.l0.b, m->road_type.ln.g, m->road_type.ln.b,
                      w.tfh, DM, rP, st));
  } else {
    for (int k = 0; k < bt.n; ++k) {
      const Cls& c = bt.c[k];
      CHK(launch_map_pool(c.B, P, d.NP, c.M, c.ctx->road_pts, m->mp, w.attn_pre + c.rP * DM, w.src_pad + c.rM, st));
      CHK(launch_in_mlp(c.ctx->road_types, 8, 8, m->road_type.l0.w, m->road_type.l0.b, m->road_type.ln.g, m->road_type.ln.b,
                        w.tfh + c.rP * DM, DM, c.B * P, st));
    }
  }
  CHK(gemm_ln(m->map_out, m->map_n1, w.attn_pre, DM, nullptr, 0, w.m1, DM, w.m1, rP, DM, 0, st));            // emb
  CHK(gemm_ln(m->map_feats.l0, m->map_feats.ln, w.m1, DM, nullptr, 0, w.m2, DM, w.m2, rP, DM, 1, st));
  CHK(gemm_ln(m->map_feats.l3, m->map_n2, w.m2, DM, w.m1, DM, w.cat, 2 * DM, w.attn_pre, rP, DM, 0, st));   // cat[:, :256]
  CHK(gemm(m->road_type.l3, w.tfh, DM, nullptr, 0, w.cat + DM, 2 * DM, rP, DM, DM, 0, st));                                                                                  // cat[:, 256:]
  CHK(gemm_ln(m->road_fuse.l0, m->road_fuse.ln, w.cat, 2 * DM, nullptr, 0, w.m2, DM, w.m2, rP, 2 * DM, 1, st));
  // final Linear: its rows ARE the polyline rows [b, 0..P-1] of the scene-encoder source — written there by the Linear's own row store
  // (weight-stationary kernel, scattered rows: round 6), or, where that kernel does not apply (bf16x6 split, f32-input family, a debug
  // caller that wants the compact rows), compact [B*P,256] + a row copy
  int scattered = 1;
  const Lin& Lf = m->road_fuse.l3;
  if (!dbg_seg_emb && Lf.w3() && ctrlsim_option(OPT_GEMM_IMPL) == 1 && ctrlsim_option(OPT_SPLIT))
    scattered = launch_gemm256_rows(w.m2, DM, Lf.w3(), Lf.ntot ? Lf.ntot : DM, Lf.n0, Lf.b, w.src, DM, w.idx_poly, rP, st);
  if (scattered < 0) return scattered;
  if (scattered == 1) {
    CHK(gemm(Lf, w.m2, DM, nullptr, 0, w.m1, DM, rP, DM, DM, 0, st));
    CHK(launch_row_copy(w.m1, DM, w.src, DM, w.idx_poly, rP, DM, 1, st));
  }
  if (dbg_seg_emb) {
    hipError_t e = hipMemcpyAsync(dbg_seg_emb, w.m1, (size_t)rP * DM * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return CTRLSIM_ELAUNCH;
  }
  // ---- scene encoder (encoder.py:155-168): post-LN layers over [polylines || initial states] with key padding
  // (the scene rows' image tails — the encoder's images and every decoder layer's memory K / V images share the geometry — in one launch)
  {
    void* sets[8];
    int ns = 0;
    sets[ns++] = w.img_enc;
    for (int i = 0; i < d.ND && ns < 8; ++i) sets[ns++] = w.img_mem[i];
    CHK(zero_kv_tails(bt, true, ns, sets, st));
  }
  const bool mem_tails_done = d.ND + 1 <= 8;
  for (int i = 0; i < d.NE; ++i) {
    const EncLayer& Le = m->enc[i];
    CHK(gemm_kv(d, bt, w, Le.qkv, w.src, w.eqkv, 3 * DM, 3 * DM, DM, w.img_enc, true, st, true));
    CHK(attention(d, bt, w, AttnCall{0, Q_SCENE, w.eqkv, 3 * DM, w.eqkv + DM, w.eqkv + 2 * DM, 3 * DM, w.img_enc, true, w.eatt, 0, 0, 0},
                  st));
    CHK(outproj_ln_ffn(Le.out, Le.n1, Le.lin1, Le.lin2, Le.n2, Le.fp, w.eatt, w.src, w.effn, w.etmp, rM, d.F, st));
  }
  // memory K/V of every decoder layer (cached for pass 2)
  for (int i = 0; i < d.ND; ++i) CHK(gemm_kv(d, bt, w, m->dec[i].ckv, w.src, w.memkv[i], 2 * DM, 2 * DM, 0, w.img_mem[i], true, st, mem_tails_done));
  return 0;
}

bool classes_ok(const ctrlsim_dims& d, const Batch& bt) {
  for (int k = 0; k < bt.n; ++k)
    if (bt.c[k].sh.rep && (d.variant != 0 || !presplit())) return false;
  return true;
}
// (round 6: the classes of a model batch in ONE launch — sixteen 7-microsecond launches in front of every forward pass were 5 664 of the
// profiled command's 76 k dispatches)
struct FillCls { int wg0, B, Areg, L, Lreg, rep_k0, M; long rL, rM; int ioff, rQ, rP, koff; };
struct FillBatch { int n; FillCls c[MAXC]; };
__global__ void fill_index_classes_kernel(FillBatch fb, int ti, int P, int qoff, int* pos_state, int* pos_rtg, int* idx_state, int* idx_rtg,
                                          int* idx_poly, int* key_all) {
  int k = 0;
  while (k + 1 < fb.n && (int)blockIdx.x >= fb.c[k + 1].wg0) ++k;
  const FillCls c = fb.c[k];
  const int i = ((int)blockIdx.x - c.wg0) * blockDim.x + threadIdx.x;
  if (i < c.B * P) idx_poly[c.rP + i] = (int)c.rM + (i / P) * c.M + (i % P);
  if (i < c.Areg) { pos_state[c.ioff + i] = (ti * c.Areg + i) * 3 + qoff; pos_rtg[c.ioff + i] = (ti * c.Areg + i) * 3 + 1; }
  if (i < c.B * c.Areg) {
    const int b = i / c.Areg, a = i - b * c.Areg;
    idx_state[c.rQ + i] = (int)c.rL + b * c.L + (ti * c.Areg + a) * 3 + qoff;
    idx_rtg[c.rQ + i] = (int)c.rL + b * c.L + (ti * c.Areg + a) * 3 + 1;
  }
  if (i < c.L) key_all[c.koff + i] = i < c.Lreg ? i : c.rep_k0 + (i - c.Lreg);
}
int launch_fill_index(const Batch& bt, const Ws& w, int P, int ti, int Tq, int qoff, hipStream_t st) {
  FillBatch fb;
  fb.n = 0;
  int wg = 0;
  for (int k = 0; k < bt.n; ++k) {
    const Cls& c = bt.c[k];
    const int nidx = max(max(c.B * c.sh.A, c.B * P), c.L);
    if (nidx <= 0) continue;
    fb.c[fb.n++] = FillCls{wg, c.B, c.sh.Areg, c.L, c.Lreg, c.sh.rep_k0(Tq), c.M, c.rL, c.rM, (int)c.ioff, (int)c.rQ, (int)c.rP, (int)c.koff};
    wg += (nidx + 255) / 256;
  }
  if (fb.n == 0) return CTRLSIM_OK;
  hipLaunchKernelGGL(fill_index_classes_kernel, dim3(wg), dim3(256), 0, st, fb, ti, P, qoff, w.pos_state, w.pos_rtg, w.idx_state, w.idx_rtg,
                     w.idx_poly, w.key_all);
  return ctrlsim_launch_status();
}
// first embedding layers: in_mlp per class (the context tensors of the classes are separate arrays), then the folded Linear once
int embed_inputs(const ctrlsim_model* m, const Batch& bt, const Ws& w, int Tn, bool goals, hipStream_t st) {
  if (bt.contig) {
    CHK(launch_in_mlp(bt.c[0].ctx->st12, 12, 12, m->embed_state.l0.w, m->embed_state.l0.b, m->embed_state.ln.g, m->embed_state.ln.b,
                      w.hS, DM, (int)bt.rS, st));
    if (goals)
      CHK(launch_in_mlp(bt.c[0].ctx->goal5, 5, 5, m->embed_goal.l0.w, m->embed_goal.l0.b, m->embed_goal.ln.g, m->embed_goal.ln.b,
                        w.hG, DM, (int)bt.rA, st));
  } else
  for (int k = 0; k < bt.n; ++k) {
    const Cls& c = bt.c[k];
    CHK(launch_in_mlp(c.ctx->st12, 12, 12, m->embed_state.l0.w, m->embed_state.l0.b, m->embed_state.ln.g, m->embed_state.ln.b,
                      w.hS + c.rS * DM, DM, c.B * Tn * c.sh.A, st));
    if (goals)
      CHK(launch_in_mlp(c.ctx->goal5, 5, 5, m->embed_goal.l0.w, m->embed_goal.l0.b, m->embed_goal.ln.g, m->embed_goal.ln.b,
                        w.hG + c.rA * DM, DM, c.B * c.sh.A, st));
  }
  CHK(gemm(m->fold_state, w.hS, DM, nullptr, 0, w.S2, DM, (int)bt.rS, DM, DM, 0, st));
  if (goals) CHK(gemm(m->fold_goal, w.hG, DM, nullptr, 0, w.Gp, DM, (int)bt.rA, DM, DM, 0, st));
  return 0;
}

}  // namespace

extern "C" int64_t ctrlsim_forward_workspace_bytes_c(const ctrlsim_dims* d, int n, const int* B, const int* A, int Tq) {
  if (!d || Tq < 1 || Tq > d->T) return CTRLSIM_EINVAL;
  Batch bt;
  ctrlsim_ctx dummy[MAXC] = {};
  if (n < 1 || n > MAXC || make_batch(*d, n, B, A, dummy, Tq, Tq, 4, bt) != CTRLSIM_OK) return CTRLSIM_EINVAL;
  return (int64_t)carve(*d, bt, nullptr).bytes;
}
extern "C" int64_t ctrlsim_forward_workspace_bytes_a(const ctrlsim_dims* d, int B, int Tq, int Actx) {
  return ctrlsim_forward_workspace_bytes_c(d, 1, &B, &Actx, Tq);
}
extern "C" int64_t ctrlsim_forward_workspace_bytes(const ctrlsim_dims* d, int B, int Tq) {
  return d ? ctrlsim_forward_workspace_bytes_a(d, B, Tq, d->A) : CTRLSIM_EINVAL;
}

// ------------------------------------------------------------------------------------------------ pass 1
namespace {
// The full forward over the first Tq window steps; logits of the head that the first pass of the variant needs, for the Areg
// regular slots of every context (rows in class order, [sum_k B_k*Areg_k, .]):
// CtRL-Sim: predict_rtg on the state tokens of the current step; IL: predict_action on the same rows; Trajeglish:
// predict_action on the action tokens (decoder.py:55-77).
struct AllOut { float *act, *rtg, *fut; };       // ctrlsim_forward_all: heads on every token, rows (b, tt, a)
int forward_full(const ctrlsim_model* m, int n, const int* Bk, const int* Ak, const ctrlsim_ctx* ctx, int Tq, void* workspace,
                 float* logits, float* dbg_seg_emb, hipStream_t st, const AllOut* all = nullptr, hipStream_t st_tail = nullptr,
                 bool split_tail = false) {
  const ctrlsim_dims& d = m->d;
  // d.variant: token layout / heads (tok_variant: 4 = CtRL-Sim tokens) and, one to one, the attention mask mode 1 + d.variant
  const int variant = tok_variant(d.variant), amode = 1 + d.variant, qoff = variant == 2 ? 2 : 0;
  if (d.variant && !presplit()) return CTRLSIM_EINVAL;     // the IL / Trajeglish / own-return masks live in the split-operand attention only
  Batch bt;
  CHK(make_batch(d, n, Bk, Ak, ctx, Tq, Tq, 4, bt));
  if (!classes_ok(d, bt)) return CTRLSIM_EINVAL;
  struct FewScope { ~FewScope() { prof_few(false); } } few_scope;    // profiling rows: full-row part, then the few-row tail
  prof_few(false);
  bt.contig = bt.n > 1 && ctx_contiguous(d, bt, Tq);
  const Ws w = carve(d, bt, static_cast<char*>(workspace));
  const int P = d.P, ti = Tq - 1, rL = (int)bt.rL, rQ = (int)bt.rQ;
  CHK(launch_fill_index(bt, w, P, ti, Tq, qoff, st));
  // ---- token embeddings (encoder.py:95-153)
  CHK(embed_inputs(m, bt, w, Tq, true, st));
  if (bt.contig) {
    int Bk[MAXC], Ak[MAXC], Ar[MAXC], Mk[MAXC];
    long xrow[MAXC], srow[MAXC], grow[MAXC];
    for (int k = 0; k < bt.n; ++k) {
      const Cls& c = bt.c[k];
      Bk[k] = c.B; Ak[k] = c.sh.A; Ar[k] = c.sh.Areg; Mk[k] = c.M; xrow[k] = c.rL; srow[k] = c.rM; grow[k] = c.rA;
    }
    CHK(launch_assemble_tokens_classes(bt.n, Bk, Ak, Ar, Mk, xrow, srow, grow, Tq, w.S2, w.Gp, bt.c[0].ctx->exist,
                                       bt.c[0].ctx->act_tok, bt.c[0].ctx->rtg_bin, bt.c[0].ctx->tstep, m->tb, w.X, w.src, P,
                                       w.src_pad, st));
  } else {
    for (int k = 0; k < bt.n; ++k) {
      const Cls& c = bt.c[k];
      CHK(launch_assemble_tokens(c.B, Tq, c.sh.A, c.sh.Areg, w.S2 + c.rS * DM, w.Gp + c.rA * DM, c.ctx->exist, c.ctx->act_tok,
                                 c.ctx->rtg_bin, c.ctx->tstep, m->tb, w.X + c.rL * DM, w.src + c.rM * DM, c.M, P, w.src_pad + c.rM, st));
    }
  }
  CHK(scene_side(m, bt, w, dbg_seg_emb, st));
  const bool use_tbl = d.variant == 0 && presplit() && ctrlsim_option(OPT_ATTN_TBL) != 0;   // (own-return mask: in-kernel masks, like the baselines)
  if (use_tbl) CHK(build_mask_tables(bt, w, Tq, st));
  // ---- decoder (decoder.py:52): layers 0..ND-2 on all tokens
  for (int i = 0; i < d.ND; ++i) {
    const DecLayer& Ld = m->dec[i];
    const bool last_few = !(i < d.ND - 1 || all);
    const bool kv_only = last_few && ctrlsim_option(OPT_LAST_KV) != 0;
    // the last layer of a rollout pass reads the queries of the A queried tokens only: keys and values of every token (two of the
    // in_proj's three column groups), queries from the gathered rows below — a twelfth of the pass's in_proj work less
    if (kv_only) CHK(gemm_kv(d, bt, w, Ld.skv, w.X, w.qkv[i] + DM, 3 * DM, 2 * DM, 0, w.img_dec[i], false, st));
    else CHK(gemm_kv(d, bt, w, Ld.qkv, w.X, w.qkv[i], 3 * DM, 3 * DM, DM, w.img_dec[i], false, st));
    if (!last_few) {
      CHK(attention(d, bt, w, AttnCall{amode, Q_ALL, w.qkv[i], 3 * DM, w.qkv[i] + DM, w.qkv[i] + 2 * DM, 3 * DM, w.img_dec[i], false,
                                       w.att, Tq, Tq, 0, use_tbl}, st));
      CHK(cross_and_ffn(m, bt, Ld, i, w, w.X, w.tmp, w.att, w.qc, w.ffn, bt.rL, Q_ALL, 0, st));
    } else {
      // last layer: only the queried tokens of the current timestep (state tokens; Trajeglish: action tokens) of the regular slots.
      // From here on every kernel touches Areg rows per context; a caller with a second stream gets this tail there, ordered
      // behind the full-row part by an event, so that the next batch's full-row kernels need not wait for it.
      if (split_tail && st_tail != st) {
        if (hipEventRecord(m->ev_tail, st) != hipSuccess || hipStreamWaitEvent(st_tail, m->ev_tail, 0) != hipSuccess) return CTRLSIM_ELAUNCH;
        st = st_tail;
      }
      prof_few(true);
      CHK(launch_row_copy(w.X, DM, w.xc, DM, w.idx_state, rQ, DM, 0, st));
      if (kv_only) CHK(gemm(Ld.sq, w.xc, DM, nullptr, 0, w.qkvc, 3 * DM, rQ, DM, DM, 0, st));        // queries of the queried rows
      else CHK(launch_row_copy(w.qkv[i], 3 * DM, w.qkvc, 3 * DM, w.idx_state, rQ, 3 * DM, 0, st));
      CHK(attention(d, bt, w, AttnCall{amode, Q_STATE, w.qkvc, 3 * DM, w.qkv[i] + DM, w.qkv[i] + 2 * DM, 3 * DM, w.img_dec[i], false,
                                       w.attc, Tq, Tq, 0}, st));
      CHK(cross_and_ffn(m, bt, Ld, i, w, w.xc, w.tmpc, w.attc, w.qcc, w.ffnc, bt.rQ, Q_STATE, 0, st));
    }
  }
  if (all) {
    // every head on every token of its type (decoder.py:55-77): token type k of (b, tt, a) is row ((b*Tq + tt)*A + a)*3 + k, so a
    // type is a strided view of X (leading dimension 3*DM).  Action head: the rtg token (CtRL-Sim), the state token (IL, and DT,
    // whose token order is rtg, state, action), the action token (Trajeglish); rtg head: state tokens; future states: action tokens.
    const int rows = rL / 3, k_act = variant == 0 ? 1 : variant == 2 ? 2 : 0;
    CHK(mlp_tail(m->head_action, w.X + k_act * DM, rows, w.att, all->act, d.V, st, 3 * DM));
    if (all->rtg) CHK(mlp_tail(m->head_rtg, w.X, rows, w.att, all->rtg, d.R * d.C, st, 3 * DM));
    if (all->fut) CHK(mlp_tail(m->head_fut, w.X + 2 * DM, rows, w.att, all->fut, 2 * d.T, st, 3 * DM));
    return CTRLSIM_OK;
  }
  // ---- predict_rtg head on the state tokens (decoder.py:74-77) / predict_action for the baselines (decoder.py:58-64)
  if (variant) return mlp_tail(m->head_action, w.xc, rQ, w.headh, logits, d.V, st);
  return mlp_tail(m->head_rtg, w.xc, rQ, w.headh, logits, d.R * d.C, st);
}
}  // namespace

extern "C" int ctrlsim_dt_forward_pass1_c(const ctrlsim_model* m, int n, const int* B, const int* A, const ctrlsim_ctx* ctx, int Tq,
                                          void* workspace, float* rtg_logits, float* dbg_seg_emb, hipStream_t st) {
  if (!m || !ctx || !workspace || !rtg_logits || Tq < 1 || Tq > m->d.T || tok_variant(m->d.variant) != 0) return CTRLSIM_EINVAL;
  return forward_full(m, n, B, A, ctx, Tq, workspace, rtg_logits, dbg_seg_emb, st);
}
// The same with the few-row tail (last decoder layer on the queried rows + the head) enqueued on `tail_stream`, behind the
// full-row part on `stream` (event-ordered inside the call).  rtg_logits are complete in tail_stream order.
extern "C" int ctrlsim_dt_forward_pass1_c2(const ctrlsim_model* m, int n, const int* B, const int* A, const ctrlsim_ctx* ctx, int Tq,
                                           void* workspace, float* rtg_logits, hipStream_t st, hipStream_t tail_stream) {
  if (!m || !ctx || !workspace || !rtg_logits || Tq < 1 || Tq > m->d.T || tok_variant(m->d.variant) != 0) return CTRLSIM_EINVAL;
  return forward_full(m, n, B, A, ctx, Tq, workspace, rtg_logits, nullptr, st, nullptr, tail_stream, true);
}
extern "C" int ctrlsim_dt_forward_pass1_a(const ctrlsim_model* m, int B, int Tq, int Actx, const ctrlsim_ctx* c, void* workspace,
                                          float* rtg_logits, float* dbg_seg_emb, hipStream_t st) {
  return ctrlsim_dt_forward_pass1_c(m, 1, &B, &Actx, c, Tq, workspace, rtg_logits, dbg_seg_emb, st);
}
extern "C" int ctrlsim_dt_forward_pass1(const ctrlsim_model* m, int B, int Tq, const ctrlsim_ctx* c, void* workspace,
                                        float* rtg_logits, float* dbg_seg_emb, hipStream_t st) {
  return m ? ctrlsim_dt_forward_pass1_a(m, B, Tq, m->d.A, c, workspace, rtg_logits, dbg_seg_emb, st) : CTRLSIM_EINVAL;
}
extern "C" int ctrlsim_dt_forward_actions(const ctrlsim_model* m, int B, int Tq, const ctrlsim_ctx* c, void* workspace,
                                          float* act_logits, hipStream_t st) {
  if (!m || !c || !workspace || !act_logits || B < 1 || Tq < 1 || Tq > m->d.T || tok_variant(m->d.variant) == 0) return CTRLSIM_EINVAL;
  const int A = m->d.A;
  return forward_full(m, 1, &B, &A, c, Tq, workspace, act_logits, nullptr, st);
}

// The reference's return contract of CtRLSim.forward (models/ctrl_sim.py:41-45, decoder.py:52-77): teacher-forced, every head on
// every token of the window.  Outputs in token-row order [B,Tq,A,.] (the reference permutes to [B,A,T,.]); rtg_preds /
// state_preds may be NULL and must be NULL for models without those heads (IL, Trajeglish, DT).  Not on the rollout path.
extern "C" int ctrlsim_forward_all(const ctrlsim_model* m, int B, int Tq, const ctrlsim_ctx* c, void* workspace, float* action_preds,
                                   float* rtg_preds, float* state_preds, hipStream_t st) {
  if (!m || !c || !workspace || !action_preds || B < 1 || Tq < 1 || Tq > m->d.T) return CTRLSIM_EINVAL;
  if ((rtg_preds && tok_variant(m->d.variant) != 0) || (state_preds && !m->has_fut)) return CTRLSIM_EINVAL;
  const int A = m->d.A;
  const AllOut all{action_preds, rtg_preds, state_preds};
  return forward_full(m, 1, &B, &A, c, Tq, workspace, nullptr, nullptr, st, &all);
}

// Component-level entry (tests, micro-benchmarks): the folded point MLP + seed-attention pooling of B*P polylines,
// modules/map_encoder.py:28-46 up to (not including) out_proj.  attn_pre [B*P,256]; pad [B,P] <- 1 for polylines without a point.
extern "C" int ctrlsim_map_pool(const ctrlsim_model* m, int B, const float* road_pts, float* attn_pre, unsigned char* pad,
                                hipStream_t st) {
  if (!m || !road_pts || !attn_pre || !pad || B < 1) return CTRLSIM_EINVAL;
  return launch_map_pool(B, m->d.P, m->d.NP, m->d.P, road_pts, m->mp, attn_pre, pad, st);
}

// ------------------------------------------------------------------------------------------------ pass 2
extern "C" int ctrlsim_dt_forward_pass2_c(const ctrlsim_model* m, int n, const int* Bk, const int* Ak, const ctrlsim_ctx* ctx, int Tq,
                                          int t, int N, int Tmax, const int* ctx_scn, const int* hist_rtg, void* workspace,
                                          float* act_logits, int cached, hipStream_t st) {
  if (!m || !ctx || !workspace || !act_logits || Tq < 1 || Tq > m->d.T || tok_variant(m->d.variant) != 0) return CTRLSIM_EINVAL;
  if (m->d.variant && (cached || !presplit())) return CTRLSIM_EINVAL;   // own-return mask: full recompute, split-operand attention only
  const ctrlsim_dims& d = m->d;
  // cached mode: the workspace is carved for the full window (K/V cache rows at rows(T) per context) and the context
  // tensors hold only the last Tn = min(Tq, 2) window rows
  const int Tw = cached ? d.T : Tq;
  const int ctx_rows = cached ? (Tq < 2 ? Tq : 2) : Tq, ti = ctx_rows - 1;
  struct FewScope { FewScope() { prof_few(true); } ~FewScope() { prof_few(false); } } few_scope;   // profiling rows: few-row launches
  Batch bt, lay;
  CHK(make_batch(d, n, Bk, Ak, ctx, Tw, ctx_rows, 4, bt));
  CHK(make_batch(d, n, Bk, Ak, ctx, Tw, cached ? 2 : ctx_rows, 4, lay));   // cached: the layout of ctrlsim_dt_forward_pass1_cached_c
  if (!classes_ok(d, bt)) return CTRLSIM_EINVAL;
  const Ws w = carve(d, lay, static_cast<char*>(workspace));
  const int rQ = (int)bt.rQ;
  int c0 = 0;
  for (int k = 0; k < bt.n; ++k) {
    const Cls& c = bt.c[k];
    CHK(launch_assemble_rtg_rows(c.B, c.sh.Areg, c.sh.A, ctx_rows, ti, t, N, Tmax, ctx_scn + c0, c.ctx->slot_gid, hist_rtg,
                                 c.ctx->exist, c.ctx->tstep, m->tb, m->zero_rtg, w.xc2 + c.rQ * DM, st));
    c0 += c.B;
  }
  for (int i = 0; i < d.ND; ++i) {
    const DecLayer& Ld = m->dec[i];
    CHK(gemm(Ld.qkv, w.xc2, DM, nullptr, 0, w.qkvc, 3 * DM, rQ, 3 * DM, DM, 0, st));
    // refresh the rtg rows' K / V: in the tile images (what the split-operand attention reads), or — f32-input MFMA family, which
    // attends over the fp32 rows themselves — in the fp32 rows of the pass-1 projection
    if (!presplit()) CHK(launch_row_copy(w.qkvc, 3 * DM, w.qkv[i], 3 * DM, w.idx_rtg, rQ, 3 * DM, 1, st));
    if (presplit()) {
      KvRowsHost kr[MAXC];
      for (int k = 0; k < bt.n; ++k) {
        const Cls& c = bt.c[k];
        kr[k] = KvRowsHost{c.B, c.sh.Areg, c.nkt_dec, c.rQ, c.tile_dec, w.pos_rtg + c.ioff};
      }
      CHK(launch_kv_split_rows_classes(w.qkvc + DM, w.qkvc + 2 * DM, 3 * DM, bt.n, kr, w.img_dec[i], st));
    }
    CHK(attention(d, bt, w, AttnCall{1 + d.variant, Q_RTG, w.qkvc, 3 * DM, w.qkv[i] + DM, w.qkv[i] + 2 * DM, 3 * DM, w.img_dec[i], false, w.attc,
                                     Tq, Tw, 0}, st));   // keys: steps <= current
    CHK(cross_and_ffn(m, bt, Ld, i, w, w.xc2, w.tmpc, w.attc, w.qcc, w.ffnc, bt.rQ, Q_RTG, 0, st));
  }
  CHK(mlp_tail(m->head_action, w.xc2, rQ, w.headh, act_logits, d.V, st));
  return CTRLSIM_OK;
}
extern "C" int ctrlsim_dt_forward_pass2_a(const ctrlsim_model* m, int B, int Tq, int Actx, int t, int N, int Tmax,
                                          const ctrlsim_ctx* c, const int* ctx_scn, const int* hist_rtg, void* workspace,
                                          float* act_logits, int cached, hipStream_t st) {
  return ctrlsim_dt_forward_pass2_c(m, 1, &B, &Actx, c, Tq, t, N, Tmax, ctx_scn, hist_rtg, workspace, act_logits, cached, st);
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
// fixed rows (rows(T) per context).  Step t only evaluates the rows whose inputs changed — the action tokens of step
// t-1 (placeholder -> applied action) and the three tokens of step t of every slot — against the cache: 4A rows instead of
// 3A*(t+1).  No other hidden state changes: an action token is visible only to later timesteps and to itself (mask closed form).
// ctx holds the window rows [max(t-1,0), t]; the workspace must be the one used at t-1 (same classes, sized with Tq = T).
extern "C" int ctrlsim_dt_forward_pass1_cached_c(const ctrlsim_model* m, int n, const int* Bk, const int* Ak, const ctrlsim_ctx* ctx,
                                                 int t, void* workspace, float* rtg_logits, hipStream_t st) {
  if (!m || !ctx || !workspace || !rtg_logits || t < 0 || t >= m->d.T || m->d.variant != 0) return CTRLSIM_EINVAL;
  const ctrlsim_dims& d = m->d;
  const int P = d.P, mul = t > 0 ? 4 : 3, tt_first = t > 0 ? t - 1 : 0, Tn = t + 1 - tt_first;
  struct FewScope { FewScope() { prof_few(true); } ~FewScope() { prof_few(false); } } few_scope;   // profiling rows: few-row launches
  Batch bt;
  CHK(make_batch(d, n, Bk, Ak, ctx, d.T, Tn, mul, bt));
  if (!classes_ok(d, bt)) return CTRLSIM_EINVAL;
  Batch bt4;                                       // the workspace layout never changes with t: carve with 4A new rows
  CHK(make_batch(d, n, Bk, Ak, ctx, d.T, 2, 4, bt4));
  Ws w = carve(d, bt4, static_cast<char*>(workspace));
  const int rQ = (int)bt.rQ, rN = (int)bt.rN;
  auto fill_cached = [&]() {
    for (int k = 0; k < bt.n; ++k) {
      const Cls& c = bt.c[k];
      const int Rn = mul * c.sh.A;
      hipLaunchKernelGGL(fill_index_cached_kernel, dim3((c.B * Rn + 255) / 256), dim3(256), 0, st, c.B, c.sh.A, c.sh.Areg, c.L, c.Lreg,
                         c.sh.rep_k0(d.T), t, Rn, c.rL, c.rN, w.pos_new + 4 * c.ioff, w.key_new + 4 * c.ioff, w.src_new + 4 * c.ioff,
                         w.idx_new + c.rN, w.idx_state_in_new + c.rQ, w.pos_rtg + c.ioff, w.idx_rtg + c.rQ);
    }
  };
  fill_cached();
  CHK(embed_inputs(m, bt, w, Tn, t == 0, st));
  if (t == 0) {
    // index lists of the Tq = 1 layout for the scene side (idx_poly); pos_rtg / idx_rtg are rewritten for the cache layout below
    for (int k = 0; k < bt.n; ++k) {
      const Cls& c = bt.c[k];
      const int nidx = max(max(c.B * c.sh.A, c.B * P), 3 * c.sh.A);
      hipLaunchKernelGGL(fill_index_kernel, dim3((nidx + 255) / 256), dim3(256), 0, st, c.B, c.sh.Areg, 3 * c.sh.A, 3 * c.sh.Areg, 0, 0, P,
                         c.M, 0, c.rN, c.rM, w.pos_state + c.ioff, w.pos_rtg + c.ioff, w.idx_state + c.rQ, w.idx_rtg + c.rQ,
                         w.idx_poly + c.rP, w.key_all + c.koff);
      // token order of assemble_tokens at Tq = 1 is the pos_new order (regular (a, k), then the representative); it also writes
      // the initial-state rows of `src`
      CHK(launch_assemble_tokens(c.B, 1, c.sh.A, c.sh.Areg, w.S2 + c.rS * DM, w.Gp + c.rA * DM, c.ctx->exist, c.ctx->act_tok,
                                 c.ctx->rtg_bin, c.ctx->tstep, m->tb, w.xn + c.rN * DM, w.src + c.rM * DM, c.M, P, w.src_pad + c.rM,
                                 st));
    }
    CHK(scene_side(m, bt, w, nullptr, st));
    fill_cached();
  } else {
    for (int k = 0; k < bt.n; ++k) {
      const Cls& c = bt.c[k];
      CHK(launch_assemble_rows(c.B, mul * c.sh.A, c.sh.A, tt_first, Tn, w.src_new + 4 * c.ioff, w.S2 + c.rS * DM, w.Gp + c.rA * DM,
                               c.ctx->exist, c.ctx->act_tok, c.ctx->rtg_bin, c.ctx->tstep, m->tb, w.xn + c.rN * DM, st));
    }
  }
  for (int i = 0; i < d.ND; ++i) {
    const DecLayer& Ld = m->dec[i];
    CHK(gemm(Ld.qkv, w.xn, DM, nullptr, 0, w.qkvn, 3 * DM, rN, 3 * DM, DM, 0, st));
    // K / V of the new rows into the cache: the tile images (split-operand attention) or the fp32 rows (f32-input MFMA family; nothing
    // reads the fp32 cache rows on the split-operand path: queries come from qkvn, keys and values from the images)
    if (!presplit()) CHK(launch_row_copy(w.qkvn, 3 * DM, w.qkv[i], 3 * DM, w.idx_new, rN, 3 * DM, 1, st));
    if (presplit()) {
      if (t == 0) {   // image tiles are read whole: stale bits beyond the written rows must at least be finite
        if (hipMemsetAsync(w.img_dec[i], 0, w.img_dec_bytes, st) != hipSuccess) return CTRLSIM_ELAUNCH;
      }
      KvRowsHost kr[MAXC];
      for (int k = 0; k < bt.n; ++k) {
        const Cls& c = bt.c[k];
        kr[k] = KvRowsHost{c.B, mul * c.sh.A, c.nkt_dec, c.rN, c.tile_dec, w.key_new + 4 * c.ioff};
      }
      CHK(launch_kv_split_rows_classes(w.qkvn + DM, w.qkvn + 2 * DM, 3 * DM, bt.n, kr, w.img_dec[i], st));
    }
    CHK(attention(d, bt, w, AttnCall{1, Q_NEW, w.qkvn, 3 * DM, w.qkv[i] + DM, w.qkv[i] + 2 * DM, 3 * DM, w.img_dec[i], false, w.attn_n,
                                     t + 1, d.T, mul}, st));
    CHK(cross_and_ffn(m, bt, Ld, i, w, w.xn, w.tmpn, w.attn_n, w.qcn, w.ffnn, bt.rN, Q_NEW, mul, st));
  }
  CHK(launch_row_copy(w.xn, DM, w.xc, DM, w.idx_state_in_new, rQ, DM, 0, st));
  CHK(mlp_tail(m->head_rtg, w.xc, rQ, w.headh, rtg_logits, d.R * d.C, st));
  return CTRLSIM_OK;
}
extern "C" int ctrlsim_dt_forward_pass1_cached_a(const ctrlsim_model* m, int B, int t, int Actx, const ctrlsim_ctx* c,
                                                 void* workspace, float* rtg_logits, hipStream_t st) {
  return ctrlsim_dt_forward_pass1_cached_c(m, 1, &B, &Actx, c, t, workspace, rtg_logits, st);
}
extern "C" int ctrlsim_dt_forward_pass1_cached(const ctrlsim_model* m, int B, int t, const ctrlsim_ctx* c, void* workspace,
                                               float* rtg_logits, hipStream_t st) {
  return m ? ctrlsim_dt_forward_pass1_cached_a(m, B, t, m->d.A, c, workspace, rtg_logits, st) : CTRLSIM_EINVAL;
}
