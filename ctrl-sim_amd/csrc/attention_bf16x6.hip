// Streaming attention with fp32-class accuracy on the 16-bit MFMA (gfx950): the split-operand variant of attention.hip.
//
// Same contract, modes, masks and work split as attention.hip (read its header first).  Difference: every fp32 operand
// of the two matrix products is split into NPL 16-bit planes and each product is evaluated as its leading partial products
// with fp32 accumulation (csrc/split.h: two fp16 planes / three products by default, three bf16 planes / six products as the
// alternative; "plane 3" / "bf16" below read NPL / the plane type).  Per 32-key sub-tile a wave issues 12 (24) 16-bit MFMAs
// = 384 (768) matrix-pipe cycles instead of 32 f32-input MFMAs (2048 cycles).
//
//   S^T = K.Q^T   A = K rows  (bf16 planes in LDS [3][k-step 2][half 2][64 keys][8]: a wave's ds_read_b128 of a fragment is
//                 two contiguous 512-byte spans, conflict-free without padding),
//                 B = the wave's Q fragment, pre-scaled by log2(e)/sqrt(32) in fp32, split once, held in registers.
//   softmax       fp32, lane-local (one query per lane column); the score MFMA chain starts from -base, so scores arrive
//                 relative to the base, and the common path exponentiates against the CURRENT base and only tests the row
//                 sums against ATT_PMAX (see the main loop); the exact-maximum / rescale path runs where that test fails.
//   O^T += V^T.P^T A = V^T (bf16 planes stored TRANSPOSED in LDS as key quads [3][16 quads][32 d][4 keys]: a wave's
//                 ds_read_b64 of a fragment half is one contiguous 256-byte span), B = P^T: the exponentiated scores of the lane are split in registers and packed in
//                 accumulator-register order.  The MFMA k-slot <-> key map is free as long as both operands agree:
//                 slot j of lane-half h  <->  key 16*kk + (j&3) + 8*(j>>2) + 4*h, which is exactly the C-fragment row
//                 map of the S^T tile, so P never moves between lanes.
// One accumulator chain per product: with three waves per SIMD the matrix pipe stays fed across the dependent MFMAs
// (measured: 1829 vs 1966 TFLOP/s register-only), and the merge adds / second rescale / 32 registers go away.
#define SPLIT_MIX   // f16x3: residual plane of the P split by v_fma_mixlo/hi_f16 (3 instructions per pair instead of 6; same bits)
#include "split.h"
#include <type_traits>

namespace SPLIT_NS {

#define ATT_PMAX 32768.0f     // row-sum bound of the speculative softmax path: every probability then fits the fp16 plane


#ifndef ATT_DIRECT_STORE
#define ATT_DIRECT_STORE 1   // 1 (round 6, default): every staged kernel stores its output rows straight from the accumulators (four 16-byte stores per
#endif                       //    lane; a lane pair completes 32 contiguous bytes, the four stores a 128-byte line) instead of transposing them through
                             //    LDS into sixteen 4-byte stores per lane: causal kernel -1.5 % (every class), rollout +0.25 % (four pairs)
#ifndef ATT_PRIO
#define ATT_PRIO 0           // 1: s_setprio 1 for waves 4-7 of the 8-wave kernels before the tile loop; 2: the softmax section at priority 1 (experiments)
#endif
#ifndef ATT_ALIGN_END
#define ATT_ALIGN_END 0      // 1: the mask-table kernel's 256-query blocks end at the last row (the partial block is the FIRST one): see the kernel
#endif
#define KT6 64
#define KV_IMG (2 * NPL * KT6 * HD)   // 16-bit elements of one (context, head, 64-key tile) image: K planes then V^T planes (8 KB per plane pair)
#define KV_PIECES (2 * NPL)          // 16-byte-per-thread LDS-DMA pieces of an image

enum { MODE6_KEYPAD = 0, MODE6_CAUSAL = 1 };

__device__ __forceinline__ opx8 cat8(const opx4 a, const opx4 b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}


// PRE = true: K / V arrive already split, as per-(context, head, 64-key tile) images of exactly the LDS stage layout
// (kv_split_kernel below; K = the image base, V / ldkv unused, kv_batch_stride = tiles per context): staging is six
// 16-byte LDS-DMA pieces per thread and tile — no registers, no VALU, no ds_write.
// One launch serves up to MAXC CLASSES of contexts (uniform shape inside a class; the engine sorts the compact contexts of a model
// batch by slot count): the 1-D grid is the concatenation of the classes' (query block, head, context) grids and a workgroup
// reads its class's shape from the table.  Offsets are in elements of the common Q / O / image / key_pad buffers.
struct AttnClass {
  long q_off, o_off, img_off, pad_off;     // class's first Q row / O row / image tile / key_pad byte
  long q_bs, o_bs, kv_bs;                  // batch strides (kv_bs: tiles per (context, head) with images, else fp32 row stride)
  const int* q_pos;
  const unsigned long long* tbl;           // TBL kernel: the class's visibility-mask table (attn_mask_table_kernel below)
  int Lq, Lk, A, rep_keys, rep_pos0, qblocks, wg0;
  float log2m;
};
// cprof (profiling runs only, else null): per slot count A two 64-bit words — shader cycles spent by the workgroups of the class (start of
// the kernel to the end of the last store, summed over workgroups) and the workgroup count (ctrlsim_attn_class_prof, bench.py)
struct AttnBatch { int n; unsigned long long* cprof; AttnClass c[MAXC]; };

// ---- visibility-mask tables (round 4) --------------------------------------------------------------------------------------------
// The structured mask (utils/train_utils.py:81-129, get_causal_mask; compact contexts: the representative's rules in the kernel header
// below) depends on the CLASS and on (query position, key position) only — not on the context, the head or the layer.  For the
// launches whose query rows are the token rows themselves (q_pos == nullptr: the first pass over all tokens, > 90 % of the attention
// time) it is therefore evaluated ONCE per forward pass into a table, in exactly the form the kernel consumes: for query group g
// (rows 32 g .. 32 g + 31 = one wave) and 32-key sub-tile j (regular sub-tiles in key order, then the representative tiles' ones) sixteen
// 64-bit LANE masks, one per accumulator register r of the S^T tile: bit (l31 + 32 half) <-> query 32 g + l31 sees key
// 32 j' + mfma_row(r, half).  The kernel loads an entry with scalar loads (wave-uniform address) and applies it with ONE
// v_cndmask_b32 per score, the SGPR pair as the lane mask — instead of building the 32-bit visibility word of every query from
// ~70 vector + 40-100 scalar instructions and applying it with two more per score.  Entry = 32 x u64: vis[16], then nob[16] = the
// representative's own tokens (multiplicity 1, where every other visible representative key counts m-fold; TBL kernel: the seed of a
// representative sub-tile always carries log2 m and these bits take it back).
#define TBL_ENTRY 32      // u64 per (query group, sub-tile)
// Round 6: the table also carries the kernel's per-sub-tile CONTROL FLOW, so that the TBL kernel derives nothing from query positions at
// run time (no divisions, no cross-lane minimum / maximum of timesteps, no block-maximum exchange through LDS + barrier, no incremental
// timestep bookkeeping per sub-tile — 6.2 scalar + 1.4 branch instructions per MFMA in round 5's counters).  Behind the mask entries
// of a class, per query group g, TBL_CTL(nkt) dwords:
//   [0] tile schedule of the 256-query block the group belongs to: n_reg | n_rep << 16 (regular tiles that hold a key some query of the
//       block sees, representative tiles likewise);  [1] the same for the group alone (32-query blocks of the streaming form);
//   [2 + tile] four bits per 64-key tile of the class's images (tile index as in the images): code of sub-tile 0 | code of sub-tile 1 << 2,
//       code 0 = no query of the group sees a key of the sub-tile: skip;  1 = every query sees every key (and none of them is one of the
//       representative's own tokens): no mask;  2 = apply the entry's visibility masks;  3 = masks + the `nob` correction.
#define TBL_CTL(nkt) (((nkt) + 3) & ~1)
#define AS4 __attribute__((address_space(4)))   // constant address space: wave-uniform loads from it are scalar loads
typedef unsigned long long u64;
typedef unsigned int u32;
struct TblClass { u64* tbl; int Lq, Lk, A, rep_keys, rep_pos0, nkt, groups, wg0; };
struct TblBatch { int n; TblClass c[MAXC]; };
__global__ __launch_bounds__(256) void attn_mask_table_kernel(TblBatch tb) {
  int ci = 0;
  while (ci + 1 < tb.n && (int)blockIdx.x >= tb.c[ci + 1].wg0) ++ci;
  const TblClass& c = tb.c[ci];
  const int g = (int)blockIdx.x - c.wg0, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int A3 = 3 * c.A, nsub = 2 * c.nkt, nsub_reg = 2 * ((c.rep_pos0 + KT6 - 1) / KT6);
  __shared__ unsigned char codes[256];                 // per sub-tile (nsub <= 2 * 64: launch_attn_mask_tables checks nkt)
  // this lane's query (rows past the end repeat the last row: such lanes must not keep a wave on the no-base-yet path)
  const int pos = min(32 * g + l31, c.Lq - 1);
  const bool rep_q = c.rep_keys > 0 && pos >= c.rep_pos0;
  int tq, aq = -1, kq;
  if (rep_q) { tq = (pos - c.rep_pos0) / 3; kq = (pos - c.rep_pos0) - 3 * tq; }
  else { tq = pos / A3; const int rem = pos - tq * A3; aq = rem / 3; kq = rem - 3 * aq; }
  u32* const ctl = reinterpret_cast<u32*>(c.tbl + (size_t)c.groups * nsub * TBL_ENTRY) + (size_t)g * TBL_CTL(c.nkt);
  if (threadIdx.x == 0) {
    // tile schedules: the largest timestep among the block's queries decides how many regular / representative keys matter
    auto sched = [&](int p0, int p1) -> u32 {            // queries at positions [p0, p1], p1 < Lq
      int bt = 0;
      const int reg_hi = min(p1, (c.rep_keys > 0 ? c.rep_pos0 : c.Lq) - 1);
      if (reg_hi >= p0) bt = reg_hi / A3;
      if (c.rep_keys > 0 && p1 >= c.rep_pos0) bt = max(bt, (p1 - c.rep_pos0) / 3);
      const int k_end = min(c.Lk, (bt + 1) * A3), rep_need = min(c.rep_keys, (bt + 1) * 3);
      return (u32)((k_end + KT6 - 1) / KT6) | ((u32)((rep_need + KT6 - 1) / KT6) << 16);
    };
    // (the kernel aligns its 256-query blocks to the END of the row range when the row count is a multiple of 32: ATT_ALIGN_END there)
    const int shift = (ATT_ALIGN_END && (c.Lq & 31) == 0) ? (256 - c.Lq % 256) % 256 : 0;
    const int b0 = ((g + (shift >> 5)) >> 3) * 256 - shift;
    ctl[0] = sched(max(b0, 0), min(b0 + 255, c.Lq - 1));
    ctl[1] = sched(32 * g < c.Lq ? 32 * g : c.Lq - 1, min(32 * g + 31, c.Lq - 1));
  }
  for (int j = wave; j < nsub; j += 4) {
    u64 mine = 0, all_v = ~0ull, any_v = 0, any_n = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = mfma_row(r, half);
      bool vis = false, nob = false;
      if (j < nsub_reg) {
        const int kp = 32 * j + kl;                  // regular key (tk, ak, kk)
        if (kp < c.Lk) {
          const int tk = kp / A3, rem = kp - tk * A3, ak = rem / 3, kk = rem - 3 * ak;
          vis = tk < tq || (tk == tq && (kk == 0 || (ak == aq && kk <= kq)));
        }
      } else {
        const int jk = 32 * (j - nsub_reg) + kl;     // representative key 3 t + k: m-fold up to the state token of the query's step
        if (jk < c.rep_keys) {
          vis = jk <= 3 * tq + (rep_q ? kq : 0);
          nob = vis && jk > 3 * tq;
        }
      }
      const u64 bv = __ballot(vis), bn = __ballot(nob);
      if (lane == r) mine = bv;
      if (lane == 16 + r) mine = bn;
      all_v &= bv; any_v |= bv; any_n |= bn;
    }
    if (lane < TBL_ENTRY) c.tbl[((size_t)g * nsub + j) * TBL_ENTRY + lane] = mine;
    // (the two sub-tiles of a tile are evaluated by different waves: the codes meet in LDS and leave as one dword per tile below)
    if (lane == 0) codes[j] = any_v == 0 ? 0 : (any_n ? 3 : (all_v == ~0ull ? 1 : 2));
  }
  __syncthreads();
  for (int t = threadIdx.x; t < c.nkt; t += 256) ctl[2 + t] = (u32)codes[2 * t] | ((u32)codes[2 * t + 1] << 2);
}

// (Tried and not kept, numbers in profiles/README.md: five workgroups per CU for the mask-table kernel; a three-stage K/V ring with the DMA
// two tiles ahead; both score products of a tile before either softmax; 256 queries per workgroup — round 5, no change at any length.)
// DIR (few-query launches: the second pass, the last decoder layer on the queried rows, the K/V-cached steps): ONE wave per workgroup
// = one 32-query group of one (context, head), and the K / V^T fragments come straight from the tile images in global memory — the image
// layout IS the fragment layout (a lane's 16 / 8 bytes are contiguous), so a tile that a single wave uses once has no business in LDS.
// The 256-thread form keeps one live wave and three idle ones per workgroup there, 33 KB of LDS each: three live waves per compute unit
// for a kernel whose whole job is to stream K / V; this form has no stage memory, no barriers that matter, ~16 waves per compute unit.
// NW = waves (32-query groups) per workgroup of the staged form.  The two full-row kernels of the rollout — causal with mask tables, key-padded
// — run 8 waves per workgroup at SIX waves per SIMD (80 VGPRs; three workgroups of 33.8 KB per CU): the kernel is latency-bound per wave
// (profiles/r05_c_pmc_attention.md: no pipe above 40 %, 3.35 waves resident per SIMD with 4 x 4), and 8-wave workgroups are what lets the LDS
// hold 24 waves per CU.  Round 5, sustained: L = 2304 2.52 -> 2.37 ms, A' = 8 1.21 -> 1.11 ms, cross 0.576 -> 0.541 ms; rollout +1.0 %
// (causal launches 2.215 -> 2.144 ms, key-padded 0.614 -> 0.594 ms).  The in-kernel-mask causal kernel (baselines, few-row non-streaming
// launches) keeps 4 waves per workgroup: at 80 registers it spills.
// QG (round 6) = 32-query groups PER WAVE of the two full-row kernels: a wave holds QG Q^T fragment sets, score tiles and output accumulators
// and reads every K / V^T fragment from LDS ONCE for all of them.  Why: tools/probes/attn_qg_probe (the loop's instruction mix on an
// LDS-resident tile, one barrier per tile, 256 queries per workgroup) sustains 97 MFMA / us / CU in the 8-wave x 32-query shape of round 5
// and 161-167 with 4 waves x 64 queries (175 = the matrix pipe at 1.4 GHz): a 32-query wave re-reads the whole 16 KB tile for 24 MFMAs —
// 128 KB of LDS reads per tile and workgroup beside 16 KB of DMA writes — and the loop was paced by the LDS, not by the pipes the counters
// of round 5 showed idle.  With QG = 2: half the LDS bytes per MFMA, half the waves at the barrier, two independent MFMA -> softmax -> MFMA
// chains in one instruction stream.  4 waves x 2 groups = the same 256 queries per workgroup, 3 workgroups per CU (168 VGPRs).
#ifndef ATT_QG
#define ATT_QG 1
#endif
constexpr int ATT_QG_FULL = ATT_QG, ATT_NW_FULL = 8 / ATT_QG;
// RES (round 6, the key-padded full-row kernel): the keys of a (context, head) are FEW — 200 polylines + the vehicles = 3.5 tiles — and every
// 256-query block used to stage them again (nine times at L = 2304) for four tiles of work between a prologue and an epilogue.  RES keeps
// ALL tiles of the (context, head) resident in LDS (NBUF = 4 stages = 66.6 KB, two workgroups per CU, four waves per SIMD): one DMA burst and ONE
// barrier per workgroup, then the workgroup walks its query blocks — Q rows in, four tiles of products, rows out (straight from the
// accumulators, 16 bytes per lane and out-quad: the LDS transpose would alias the resident tiles) — with no barrier, no DMA and no wait
// in the loop.  L2 -> LDS traffic of the launch: one image read per (context, head) instead of one per query block.
template <int MODE, bool PRE, bool TBL = false, bool DIR = false, int NW = 4, int QG = 1, bool RES = false>
#ifndef ATT_V_EARLY
#define ATT_V_EARLY 0     // 1 (experiment): the full-row kernels request the V^T fragments BEFORE the softmax, at five waves per SIMD (96 VGPRs)
#endif
__global__ __launch_bounds__(DIR ? 64 : 64 * NW, DIR ? 3 : (RES ? 4 : (NW == 8 ? (ATT_V_EARLY ? 5 : 6) : 3))) void attention_bf16x6_kernel(
    const float* __restrict__ Qb_, int ldq, const float* __restrict__ K, const float* __restrict__ V, int ldkv,
    float* __restrict__ Ob_, int ldo, const unsigned char* __restrict__ key_pad_, float scale_log2e, int variant, AttnBatch ab) {
  const unsigned long long t_start = ab.cprof ? __builtin_amdgcn_s_memtime() : 0ull;
  int ci = 0;
  while (ci + 1 < ab.n && (int)blockIdx.x >= ab.c[ci + 1].wg0) ++ci;
  const AttnClass& cd = ab.c[ci];
  const int Lq = cd.Lq, Lk = cd.Lk, A = cd.A, rep_keys = cd.rep_keys, rep_pos0 = cd.rep_pos0, nqb = cd.qblocks;
  const float log2m = cd.log2m;
  const long q_batch_stride = cd.q_bs, o_batch_stride = cd.o_bs, kv_batch_stride = cd.kv_bs;
  const int* __restrict__ q_pos = cd.q_pos;
  const float* __restrict__ Q = Qb_ + cd.q_off;
  float* __restrict__ O = Ob_ + cd.o_off;
  const unsigned char* __restrict__ key_pad = key_pad_ ? key_pad_ + cd.pad_off : nullptr;
  // rep_keys > 0 (causal, PRE): COMPACT contexts.  Token rows of agent slots that never exist in the window are all equal
  // (every embedding is multiplied by the existence flag before embed_ln, modules/encoder.py:127-133, and the decoder has
  // no key padding on its targets), and by induction over the layers so are their hidden states at equal (timestep, token
  // type): the 24 - n padded slots of a context are therefore evaluated ONCE, as a "representative" slot whose tokens stand
  // for m = 2^log2m identical ones.  Sequence order of such a context: Lk "regular" tokens, (timestep, slot < A, type) as
  // always (positions < rep_pos0 = the regular length of the FULL window; Lk of them are attended), then rep_keys = 3 Tq
  // representative tokens (timestep, type) at positions rep_pos0 + 3 t + k, whose K/V tiles start at tile ceil(rep_pos0 / 64).
  // A representative key (t, k) is visible to a query of timestep tq iff t < tq or (t == tq and k == 0), each time with
  // multiplicity m (+log2m on the score: softmax over m equal keys); to the representative's own queries (positions
  // >= rep_pos0) also its own tokens 1..kq of step tq, once.  Representative queries see regular keys like a query without own
  // tokens (earlier steps, and the state tokens of their step).
  constexpr int K_PLANE = KT6 * HD;              // bf16 elements: [ks 2][half 2][64 keys][8]
  constexpr int V_PLANE = HD * KT6;              // [16 quads][32 d][4 keys]
  constexpr int BUF = NPL * (K_PLANE + V_PLANE) + 2 * KT6 + 8;   // + KT6 floats of key-padding bias + two "sub-tile has padded keys" flags
  // Stage ring of the staged form (PRE, not DIR): NBUF buffers, the LDS-DMA of tile it + NBUF - 1 is requested at the top of tile it.
  // NBUF = 3 (-DATT_NBUF_FULL=3; round 6, the 8-wave full-row kernels: 3 x 16.6 KB x three workgroups = 150 KB of the CU's 160 KB at the SAME
  // six waves per SIMD — the round-3/4 trial of a three-stage ring had cost a workgroup of occupancy): a tile's image has two tiles of
  // compute to land instead of one.  Still ONE barrier per tile: it publishes tile it + 1 and retires buffer it % NBUF, which the request
  // at the top of tile it + 1 (for tile it + NBUF) refills.  MEASURED (tools/jobs/r06_c.sh, sustained, same box): no change at any length
  // (L = 2304 2.356 -> 2.357 ms, compact classes and cross attention within 0.5 %) although the kernel WITHOUT its image DMA
  // (-DATT_ABL_NODMA, wrong results) runs 16 % (causal) / 22 % (cross) faster: what the DMA costs is not exposed latency.  Default 2.
#ifndef ATT_NBUF_FULL
#define ATT_NBUF_FULL 2
#endif
  constexpr int NBUF = RES ? 4 : ((PRE && !DIR && NW == 8) ? ATT_NBUF_FULL : 2);
  static_assert(!RES || (MODE == MODE6_KEYPAD && PRE && !DIR && !TBL && QG == 1), "resident keys: the key-padded staged kernel");
  static_assert(!DIR || PRE, "the streaming form reads pre-split images");
  constexpr int PADSZ = 2 * KT6 + 8;             // 16-bit elements of a stage's key-padding bias block
  constexpr int BUF_ = DIR ? PADSZ : BUF, NBUF_ = NBUF;
  static_assert(NW == 4 || (PRE && !DIR), "more than four waves per workgroup: staged pre-split images only");
  static_assert(QG == 1 || (PRE && !DIR && (TBL || MODE == MODE6_KEYPAD)), "several query groups per wave: the two full-row kernels only");
  constexpr int ARENA_T = NW * QG * 32 * 33 * 2;                          // 16-bit elements of the output transpose (NW x QG x 32 x 33 floats)
  constexpr int ARENA = DIR ? 2 * PADSZ + 32 * 33 * 2 : (NBUF_ * BUF_ > ARENA_T ? NBUF_ * BUF_ : ARENA_T);     // DIR: two bias blocks, then the 32 x 33 floats of the output transpose
  __shared__ int blk_tmax[NW];

  // XCD-aware work map: workgroups are dealt round-robin to the 8 XCDs (linear id % 8) and each XCD has its own L2, so all
  // query blocks of one (context, head) — which re-read the same K/V tiles — are given to ONE XCD: head = linear id % 8.
  // (With the natural x-fastest order the 18 query blocks of a head were spread over all 8 L2s: measured 2.8x the
  // algorithmic HBM-side fetch traffic.)  Causal: longest query blocks first.
  const int lin = (int)blockIdx.x - cd.wg0;
  const int h = lin & (NHEAD - 1), j = lin >> 3;
  const int qx = j % nqb, b = j / nqb;
  // RES: nqb = workgroups per (context, head); workgroup qx walks the query blocks [q_lo, q_hi) of the nqb_total the class has
  const int nqb_total = (Lq + 32 * NW * QG - 1) / (32 * NW * QG);
  const int q_lo = RES ? qx * nqb_total / nqb : ((MODE == MODE6_CAUSAL) ? (nqb - 1 - qx) : qx);
  const int q_hi = RES ? (qx + 1) * nqb_total / nqb : q_lo + 1;
  const int tid = threadIdx.x, wave = DIR ? 0 : tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int A3 = 3 * A;
  const float NEG_INF = -__builtin_inff();

  // ---- tile 0's K/V image pieces are requested before anything else (PRE: every query block needs tile 0; the request needs only
  // (context, head)): the L2 / HBM latency of the first tile then runs underneath the Q loads and the mask bookkeeping
  const op_t* img = PRE ? reinterpret_cast<const op_t*>(K) + (cd.img_off + ((size_t)b * NHEAD + h) * (size_t)cd.kv_bs) * KV_IMG : nullptr;
  __shared__ __attribute__((aligned(16))) op_t arena[ARENA];
  auto dma_tile = [&](int tile, int buf) {
    if (DIR) return;
#ifdef ATT_ABL_NODMA
    return;                                                              // ablation: stale LDS contents (wrong results), no image traffic
#endif
    const op_t* src = img + (size_t)tile * KV_IMG + tid * 8;
#pragma unroll
    for (int i = 0; i < KV_PIECES * 4 / NW; ++i) {
      op_t* dst = arena + buf * BUF_ + (wave * 64 + 64 * NW * i) * 8;      // wave-uniform LDS base (+ 16 B per lane)
      // Issued as inline asm on purpose: hipcc answers the builtin with `s_waitcnt vmcnt(0)` in front of the NEXT ds_read
      // of any LDS address (it cannot tell the two stage buffers apart), i.e. every wave sat out the whole L2 / HBM
      // latency of the tile it had just requested before touching the tile it already had.  The compiler does not see
      // this load; the wave waits for its own pieces explicitly right before the end-of-tile barrier (dma_wait).
      const unsigned lds_addr = (unsigned)(size_t)((__attribute__((address_space(3))) op_t*)dst);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off ; KV-DMA"
                   :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(src + 64 * NW * 8 * i) : "memory");
    }
  };
  if (PRE && !RES) dma_tile(0, 0);
  if (RES) {
    // every tile of the (context, head) + its key-padding bias block, once; ceil(Lk / 64) <= NBUF is the launcher's condition for this kernel
    const int nt = (Lk + KT6 - 1) / KT6;
    for (int t = 0; t < nt; ++t) {
      dma_tile(t, t);
      if (tid < KT6) {
        const int kr = t * KT6 + tid;
        const float pp = (kr < Lk && !key_pad[(size_t)b * Lk + kr]) ? 0.f : NEG_INF;
        const unsigned long long bal = __ballot(pp != 0.f);
        float* pb_ = reinterpret_cast<float*>(arena + t * BUF_ + NPL * (K_PLANE + V_PLANE));
        pb_[tid] = pp;
        if (tid < 2) reinterpret_cast<int*>(pb_ + KT6)[tid] = (unsigned)(bal >> (32 * tid)) != 0u;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) ; KV-DMA landed (all tiles)" ::: "memory");
    __syncthreads();
  }
  // RES: the Q rows of the NEXT query block are requested while the current one is computed (a wave has nothing else in flight in its loop)
  f32x4 qraw[RES ? 4 : 1];
  auto q_request = [&](int qblk_) {
    const int row = min(qblk_ * (32 * NW * QG) + wave * QG * 32 + l31, Lq - 1);
    const float* qp = Q + (size_t)b * q_batch_stride + (size_t)row * ldq + h * HD + half * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qraw[RES ? 2 * ks : 0] = *reinterpret_cast<const f32x4*>(qp + ks * 16);
      qraw[RES ? 2 * ks + 1 : 0] = *reinterpret_cast<const f32x4*>(qp + ks * 16 + 4);
    }
  };
  if (RES && q_lo < q_hi) q_request(q_lo);
  for (int qblk = q_lo; qblk < q_hi; ++qblk) {          // (one query block per workgroup unless RES)
  // TBL (round 6): the query blocks are aligned to the END of the row range — block k = rows [256 k - shift, 256 k - shift + 256), shift = the
  // padding of the row count to a multiple of 256 — so that the PARTIAL block is the first one, whose queries see the fewest keys (3-5
  // tiles), instead of the last one (the whole key range: up to 38 tiles with half or more of the workgroup's waves dead through all of them:
  // L = 1056 left seven of eight waves idle for 19 tiles).  Groups stay 32-aligned (shift is a multiple of 32 when the row count is).
  // MEASURED (tools/jobs/r06_l.sh, sustained, same box; -DATT_ALIGN_END=1): A' = 11 / 6 gain 4 / 2.5 %, A' = 4 / 9 / 14 / 20 LOSE 5 / 4 / 3.5 / 1 %,
  // rollout 143.7 -> 143.3 k (-0.3 %, two pairs): dead waves do not hold the matrix pipe, and the full last blocks now all finish together.
  // Off (default 0); tokens identical either way.
  const int qshift = (ATT_ALIGN_END && TBL && !DIR && (Lq & 31) == 0) ? (32 * NW * QG - Lq % (32 * NW * QG)) % (32 * NW * QG) : 0;
  const int qb = qblk * (DIR ? 32 : 32 * NW * QG) - qshift;

  // ---- this lane's query: fragment of Q^T (B operand), k-step ks covers d = 16*ks + 8*half .. +7
  // (QG > 1: group g of this wave = queries qb + (wave * QG + g) * 32 + l31; everything per-query below is an array over g)
  const int qi = qb + wave * QG * 32 + l31;
  const bool qvalid = qi >= 0 && qi < Lq;
  // a wave whose 32 query slots all lie beyond Lq (few-query calls: the second pass, the last decoder layer, the K/V-cached
  // steps fill one wave of the four) stages tiles and keeps the barriers, but skips the products and the softmax
  bool glive[QG];                                    // group g has queries (wave-uniform): its first row lies in [0, Lq)
  bool wave_live = false;
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    glive[g] = (unsigned)__builtin_amdgcn_readfirstlane(qb + (wave * QG + g) * 32) < (unsigned)Lq;
    wave_live = wave_live || glive[g];
  }
  const int qrow = qvalid ? qi : (qi < 0 ? 0 : Lq - 1);
  const int pos = q_pos ? q_pos[qrow] : qrow;
  int tq = 0, aq = 0, kq = 0;
  bool rep_q = false;                               // this lane's query is a representative token
  if (MODE == MODE6_CAUSAL && !TBL) {               // (TBL: masks, skips and the tile schedule all come from the class table)
    rep_q = rep_keys > 0 && pos >= rep_pos0;
    if (rep_q) {
      tq = (pos - rep_pos0) / 3;
      kq = (pos - rep_pos0) - tq * 3;
    } else {
      tq = pos / A3;
      const int rem = pos - tq * A3;
      aq = rem / 3;
      kq = rem - aq * 3;
    }
  }
  opx8 qf[QG][2][NPL];
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    const int qrow_g = g == 0 ? qrow : max(min(qi + 32 * g, Lq - 1), 0);
    const float* qp = Q + (size_t)b * q_batch_stride + (size_t)qrow_g * ldq + h * HD + half * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 x0 = RES ? qraw[RES ? 2 * ks : 0] : *reinterpret_cast<const f32x4*>(qp + ks * 16);
      f32x4 x1 = RES ? qraw[RES ? 2 * ks + 1 : 0] : *reinterpret_cast<const f32x4*>(qp + ks * 16 + 4);
      x0 *= scale_log2e;
      x1 *= scale_log2e;
      const float xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      split_frag(xs, qf[g][ks]);
    }
  }
  if (RES && qblk + 1 < q_hi) q_request(qblk + 1);

  // ---- key range
  int k_end = Lk, rep_need = 0;                    // regular keys [0, k_end) and representative keys [0, rep_need) matter
  int tq_min_w = 0, tq_max_w = 0;
  // TBL: this wave's query group in the class's mask table, and the group's control words behind the class's mask entries
  static_assert(!TBL || DIR || NW * QG == 8, "the table's block schedules (TBL_CTL word 0) are those of 256-query blocks");
  const int nsub_tbl = 2 * (int)kv_batch_stride;
  int qgrp[QG];
  const AS4 u32* ctl[QG];
  int tbl_n_reg = 0, tbl_n_rep = 0;
#pragma unroll
  for (int g = 0; g < QG; ++g) { qgrp[g] = __builtin_amdgcn_readfirstlane(DIR ? qblk : (qblk * NW + wave) * QG + g - (qshift >> 5)); ctl[g] = nullptr; }
  if (TBL) {
    const int groups = 4 * ((Lq + 127) / 128), live = (Lq + 31) >> 5;
    // (a wave without queries — the tail of the last 256-query block — still stages tiles: it takes the schedule of the block's last live group)
#pragma unroll
    for (int g = 0; g < QG; ++g)
      ctl[g] = (const AS4 u32*)(cd.tbl + (size_t)groups * nsub_tbl * TBL_ENTRY) + (size_t)max(min(qgrp[g], live - 1), 0) * TBL_CTL((int)kv_batch_stride);
    const u32 hdr = ctl[0][DIR ? 1 : 0];
    tbl_n_reg = (int)(hdr & 0xffffu); tbl_n_rep = (int)(hdr >> 16);
  }
  if (MODE == MODE6_CAUSAL && !TBL) {
    int tmin = tq, tmax = tq;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      tmin = min(tmin, __shfl_xor(tmin, o, 64));
      tmax = max(tmax, __shfl_xor(tmax, o, 64));
    }
    tq_min_w = __builtin_amdgcn_readfirstlane(tmin);       // wave-uniform: keep them (and the branches on them) scalar
    tq_max_w = __builtin_amdgcn_readfirstlane(tmax);
    if (lane == 0) blk_tmax[wave] = tmax;
    __syncthreads();
    int bt = tq_max_w;
    if (!DIR) {
#pragma unroll
      for (int k = 0; k < NW; ++k) bt = max(bt, blk_tmax[k]);
    }
    k_end = __builtin_amdgcn_readfirstlane(min(Lk, (bt + 1) * A3));
    rep_need = __builtin_amdgcn_readfirstlane(min(rep_keys, (bt + 1) * 3));
  }


  f32x16 oa[QG];                                   // O^T accumulators
#pragma unroll
  for (int g = 0; g < QG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) oa[g][r] = 0.f;
  int tk_t = 0, tk_r = 0;                          // (timestep, offset in it) of the next sub-tile's first key
  const int t_last = (MODE == MODE6_CAUSAL) ? (Lk - 1) / A3 : 0;
  float m_run[QG], l_run[QG];
#pragma unroll
  for (int g = 0; g < QG; ++g) { m_run[g] = NEG_INF; l_run[g] = 0.f; }

  const float* Kb = K + (size_t)b * kv_batch_stride + h * HD;
  const float* Vb = V + (size_t)b * kv_batch_stride + h * HD;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // staging registers: K rows (idx -> row, 4 consecutive d), V row PAIRS (thread -> keys 2rp, 2rp+1, 4 consecutive d)
  f32x4 pk[2], pv[2];
  float ppad = 0.f;
  unsigned long long ppad_bal = ~0ull;             // wave 0: which of the staged tile's 64 keys are padded
  auto gload = [&](int k0, int buf, bool dma = true) {
    if (PRE) {
      if (dma) dma_tile(k0 / KT6, buf);
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, r = idx >> 3, c = (idx & 7) * 4;
      const int kr = k0 + r;
      pk[i] = kr < Lk ? *reinterpret_cast<const f32x4*>(Kb + (size_t)kr * ldkv + c) : zero4;
      const int vr = k0 + 2 * (tid >> 3) + i;
      pv[i] = vr < Lk ? *reinterpret_cast<const f32x4*>(Vb + (size_t)vr * ldkv + (tid & 7) * 4) : zero4;
    }
    }
    if (MODE == MODE6_KEYPAD && tid < KT6) {
      const int kr = k0 + tid;
      ppad = (kr < Lk && !key_pad[(size_t)b * Lk + kr]) ? 0.f : NEG_INF;
      ppad_bal = __ballot(ppad != 0.f);
    }
  };
  auto sstore = [&](int buf) {
    op_t* Kd = arena + buf * BUF;
    op_t* Vd = Kd + NPL * K_PLANE;
    if (!PRE) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, r = idx >> 3, c = (idx & 7) * 4;
      u32x2 kp[NPL];
      split_quad(pk[i], kp);                                      // pairs along d: the fragment order
      const int ko = ((c >> 3) * KT6 + r) * 8 + (c & 7);          // (ks*2 + half) = c >> 3
#pragma unroll
      for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(Kd + q * K_PLANE + ko) = kp[q];
    }
    {  // V^T: pairs along the key axis (keys 2rp, 2rp+1 of one d) -> one 32-bit store per (plane, d)
      const int rp = tid >> 3, c = (tid & 7) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned vp[NPL];
        split_pair(pv[0][e], pv[1][e], vp);
        const int vo = ((rp >> 1) * HD + (c + e)) * 4 + (rp & 1) * 2;   // [quad][d][key & 3]
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<unsigned*>(Vd + q * V_PLANE + vo) = vp[q];
      }
    }
    }
    if (MODE == MODE6_KEYPAD && tid < KT6) {
      float* pb_ = DIR ? reinterpret_cast<float*>(arena + buf * PADSZ) : reinterpret_cast<float*>(Vd + NPL * V_PLANE);
      pb_[tid] = ppad;
      // a 32-key sub-tile without a padded key (the rule: every polyline and vehicle row of a scene is a valid key) skips the bias
      // reads and adds in the loop below
      if (tid < 2) reinterpret_cast<int*>(pb_ + KT6)[tid] = (unsigned)(ppad_bal >> (32 * tid)) != 0u;
    }
  };

  // ---- tile schedule: the regular tiles [0, n_reg) that hold keys < k_end, then the representative tiles (compact contexts)
  const int n_reg = TBL ? tbl_n_reg : (k_end + KT6 - 1) / KT6, n_rep = TBL ? tbl_n_rep : (rep_need + KT6 - 1) / KT6, n_it = n_reg + n_rep;
  const int nkt_reg = (rep_pos0 + KT6 - 1) / KT6;
  auto tile_k0 = [&](int it) { return (it < n_reg ? it : nkt_reg + (it - n_reg)) * KT6; };
  if (!RES) {
  if (n_it > 0) {
    gload(tile_k0(0), 0, false);                 // PRE: tile 0 (always a regular tile: k_end >= 1) was requested at the top
    sstore(0);
  }
  if (PRE && !DIR) asm volatile("s_waitcnt vmcnt(0) ; KV-DMA landed" ::: "memory");
  __syncthreads();
  }
  if (PRE && !DIR && !RES) {
#pragma unroll
    for (int a = 1; a < NBUF - 1; ++a)                                  // NBUF = 3: tile 1 is requested here, tile it + 2 at the top of tile it
      if (a < n_it) dma_tile(tile_k0(a) / KT6, a);
  }

  // DIR: the K and V^T fragments of sub-tile i + 1 are requested (global loads into a second register set) before sub-tile i is
  // computed — one wave has nothing else to cover the two memory round trips per sub-tile with (272 us per launch in the K/V-cached
  // phase without it).  The sub-tile after the last one re-reads the last one (never used).
  opx8 nk0[NPL], nk1[NPL], nv0[NPL], nv1[NPL];
  auto dir_fetch = [&](int it_, int sub_) {
    const op_t* Ks_ = img + (size_t)(tile_k0(it_) / KT6) * KV_IMG;
    const op_t* kr_ = Ks_ + (half * KT6 + sub_ * 32 + l31) * 8;
    const op_t* vr_ = Ks_ + NPL * K_PLANE + ((sub_ * 8 + half) * HD + l31) * 4;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      nk0[p] = *reinterpret_cast<const opx8*>(kr_ + p * K_PLANE);
      nk1[p] = *reinterpret_cast<const opx8*>(kr_ + p * K_PLANE + 2 * KT6 * 8);
      nv0[p] = cat8(*reinterpret_cast<const opx4*>(vr_ + p * V_PLANE), *reinterpret_cast<const opx4*>(vr_ + p * V_PLANE + 2 * HD * 4));
      nv1[p] = cat8(*reinterpret_cast<const opx4*>(vr_ + p * V_PLANE + 4 * HD * 4), *reinterpret_cast<const opx4*>(vr_ + p * V_PLANE + 6 * HD * 4));
    }
  };
  if (DIR && n_it > 0) dir_fetch(0, 0);
  int cur = 0;
#if ATT_PRIO == 1
  // round-6 experiment (MI355X_MICROARCH.md, "static priority for the younger half"): the second-dispatched half of an 8-wave workgroup loses
  // the VALU arbitration on every segment; one s_setprio before the loop, no per-segment flips
  if (NW == 8 && !DIR && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  for (int it = 0; it < n_it; ++it, cur = (cur + 1 == NBUF ? 0 : cur + 1)) {
    const bool more = it + 1 < n_it;
    const int nxt = cur + 1 == NBUF ? 0 : cur + 1;                       // buffer of tile it + 1
    const bool ahead = (PRE && !DIR) && it + NBUF - 1 < n_it;            // a tile to request now (tile it + NBUF - 1)
    if (RES) {
      // (every tile is resident in buffer `it`: nothing to request, nothing to wait for)
    } else if (PRE && !DIR) {
      if (ahead) dma_tile(tile_k0(it + NBUF - 1) / KT6, cur == 0 ? NBUF - 1 : cur - 1);
      if (more) gload(tile_k0(it + 1), nxt, false);                      // (key-padding bias of the NEXT tile: registers now, LDS at the end)
    } else if (more) gload(tile_k0(it + 1), nxt);
    const bool rep_tile = it >= n_reg;               // wave-uniform: a tile of representative keys (compact contexts)
    const int k0 = it * KT6;
    u32 cw[QG];                                      // TBL: this tile's control words (requested here, first used behind the DMA issue above)
#pragma unroll
    for (int g = 0; g < QG; ++g) cw[g] = (TBL && glive[g]) ? ctl[g][2 + tile_k0(it) / KT6] : 0u;
    // DIR: the fragments are read from the tile image itself (same layout as a stage: the DMA copies images verbatim)
    const op_t* Ks = DIR ? img + (size_t)(tile_k0(it) / KT6) * KV_IMG : arena + cur * BUF;
    const op_t* Vs = Ks + NPL * K_PLANE;
    const float* padbias = DIR ? reinterpret_cast<const float*>(arena + cur * PADSZ) : reinterpret_cast<const float*>(Vs + NPL * V_PLANE);
    int padflag[2] = {1, 1};
    if (MODE == MODE6_KEYPAD) {
      padflag[0] = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(padbias + KT6)[0]);
      padflag[1] = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(padbias + KT6)[1]);
    }

#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      opx8 ck0[NPL], ck1[NPL], cv0[NPL], cv1[NPL];
      if (DIR) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) { ck0[p] = nk0[p]; ck1[p] = nk1[p]; cv0[p] = nv0[p]; cv1[p] = nv1[p]; }
        const bool last_sub = sub == 1 && it + 1 >= n_it;
        dir_fetch(sub == 0 || last_sub ? it : it + 1, last_sub ? 1 : sub ^ 1);
      }
      if (!wave_live) continue;
      const int ks0 = k0 + sub * 32;
      int t_lo = 0, t_hi = 0, ks_t0 = 0;
      const int j0 = (it - n_reg) * KT6 + sub * 32;   // first representative key of this sub-tile (rep_tile)
      // per query group of this wave: does it take part in this sub-tile (act), with masks (need_mask); TBL: the table's code
      bool act[QG], need_mask[QG];
      u32 code[QG];
#pragma unroll
      for (int g = 0; g < QG; ++g) { act[g] = glive[g]; need_mask[g] = true; code[g] = (cw[g] >> (2 * sub)) & 3u; }
      if (TBL) {
#pragma unroll
        for (int g = 0; g < QG; ++g) {
          act[g] = code[g] != 0;                      // 0: no query of the group sees a key of the sub-tile (dead groups: cw = 0)
          need_mask[g] = code[g] >= 2;
        }
      } else if (!rep_tile) {
        // timestep of the first / last key of this sub-tile (tracked incrementally: no divisions in the loop)
        t_lo = tk_t;
        t_hi = min(tk_t + (tk_r + 31 >= A3 ? (tk_r + 31 - A3 >= A3 ? (tk_r + 31) / A3 : 1) : 0), t_last);
        ks_t0 = ks0 - tk_r;                           // position of the first key of timestep t_lo
        tk_r += 32;
        while (tk_r >= A3) { tk_r -= A3; ++tk_t; }
        if (ks0 >= k_end) continue;
        if (MODE == MODE6_CAUSAL) {
          if (t_lo > tq_max_w) continue;
          need_mask[0] = !(t_hi < tq_min_w && ks0 + 31 < Lk) || variant != 0;   // IL / Trajeglish: dead key types everywhere
        }
      } else if (j0 >= rep_need || j0 > 3 * tq_max_w + 2) {
        continue;
      } else {
        // representative keys of steps before every query of this wave: all 32 visible, all m-fold -> no mask, and the
        // multiplicity enters through the accumulator seed below
        need_mask[0] = !(j0 + 31 <= 3 * tq_min_w && j0 + 32 <= rep_keys);
      }
      if (QG > 1) {
        bool any = false;
#pragma unroll
        for (int g = 0; g < QG; ++g) any = any || act[g];
        if (!any) continue;
      } else if (!act[0]) {
        continue;
      }
      // ---- K fragments of the sub-tile: read from LDS ONCE for all query groups of the wave
      opx8 k0f[NPL], k1f[NPL];
      {
        const op_t* kr_ = Ks + (half * KT6 + sub * 32 + l31) * 8;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          if (DIR) { k0f[p] = ck0[p]; k1f[p] = ck1[p]; continue; }
          k0f[p] = *reinterpret_cast<const opx8*>(kr_ + p * K_PLANE);                  // k-step 0 (d 0-15)
          k1f[p] = *reinterpret_cast<const opx8*>(kr_ + p * K_PLANE + 2 * KT6 * 8);    // k-step 1 (d 16-31)
        }
      }
      // ---- S^T = K . Q^T per group: one accumulator chain, k-steps d 0-15 and d 16-31, six partial products each
      // the accumulator starts at -m_base (the running maximum, 0 before the first visible key): the MFMA chain then
      // delivers S - m directly and the per-element subtraction is needed only in the (rare) sub-tiles that raise the maximum
      f32x16 s0[QG];
      u64 mk[QG][16];
      const AS4 u64* tbl_e[QG];
      float m_base[QG];
#pragma unroll
      for (int g = 0; g < QG; ++g) {
        tbl_e[g] = nullptr;
        m_base[g] = (m_run[g] == NEG_INF) ? 0.f : m_run[g];
        if (!act[g]) continue;
        // TBL: every representative sub-tile is seeded with the multiplicity; the table's `nob` bits take it back where a key counts once
        const float seed = (rep_tile && (TBL || !need_mask[g])) ? log2m - m_base[g] : -m_base[g];
        // TBL: this sub-tile's lane masks, requested (scalar loads, wave-uniform address) before the score products
        if (TBL && MODE == MODE6_CAUSAL && need_mask[g]) {
          const int jsub = (rep_tile ? nkt_reg + (it - n_reg) : it) * 2 + sub;
          tbl_e[g] = (const AS4 u64*)(cd.tbl) + ((size_t)qgrp[g] * nsub_tbl + jsub) * TBL_ENTRY;
#pragma unroll
          for (int r = 0; r < 16; ++r) mk[g][r] = tbl_e[g][r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[g][r] = seed;
#define QK(PA, PB)                                \
  s0[g] = MFMA_OP(k0f[PA], qf[g][0][PB], s0[g]);  \
  s0[g] = MFMA_OP(k1f[PA], qf[g][1][PB], s0[g]);
        PROD_LIST(QK)
#undef QK
      }
      // ---- V^T fragments: A = V^T rows d = l31, slots 0-3 <-> keys 16kk+4half+0..3, slots 4-7 <-> +8; read ONCE for all groups.
      // QG > 1: requested here, their LDS latency runs underneath the score chains; QG = 1: right in front of the P^T . V^T products as in
      // round 5 (at 80 registers the early request spills)
      opx8 v0f[NPL], v1f[NPL];
      auto load_v = [&]() {
        // quad of subtile-local keys [4q', 4q'+3] is quad index sub*8 + q'; lane half h needs q' = 4kk + h and 4kk + 2 + h
        const op_t* vr_ = Vs + ((sub * 8 + half) * HD + l31) * 4;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          if (DIR) { v0f[p] = cv0[p]; v1f[p] = cv1[p]; continue; }
          const opx4 a0 = *reinterpret_cast<const opx4*>(vr_ + p * V_PLANE);
          const opx4 a1 = *reinterpret_cast<const opx4*>(vr_ + p * V_PLANE + 2 * HD * 4);
          const opx4 b0 = *reinterpret_cast<const opx4*>(vr_ + p * V_PLANE + 4 * HD * 4);
          const opx4 b1 = *reinterpret_cast<const opx4*>(vr_ + p * V_PLANE + 6 * HD * 4);
          v0f[p] = cat8(a0, a1);
          v1f[p] = cat8(b0, b1);
        }
      };
      constexpr bool V_EARLY = QG > 1 || (ATT_V_EARLY && NW == 8 && !RES);
      if (V_EARLY) load_v();
#pragma unroll
      for (int g = 0; g < QG; ++g) {
      if (!act[g]) continue;
      float sc[16];
      if (MODE == MODE6_KEYPAD && padflag[sub]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = s0[g][r] + padbias[sub * 32 + mfma_row(r, half)];
      } else if (TBL && MODE == MODE6_CAUSAL && need_mask[g]) {
        // one v_cndmask_b32 per score, the table's SGPR pair as the lane mask.  (Through the inverse-ballot builtin, NOT inline asm: a
        // hand-written VALU instruction that reads the accumulator of the MFMA just issued gets no wait states from hipcc's hazard
        // recognizer and reads the registers before the matrix pipe has written them — the first version of this path did.)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_inverse_ballot_w64(mk[g][r]) ? s0[g][r] : NEG_INF;
        if (code[g] == 3) {
          // the representative's own tokens of its step count once: take the multiplicity back (three query groups per context only)
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[r] -= __builtin_amdgcn_inverse_ballot_w64(tbl_e[g][16 + r]) ? log2m : 0.f;
        }
      } else if (MODE == MODE6_CAUSAL && need_mask[g]) {
        // visibility of the 32 keys of this sub-tile for this lane's query as a bit mask (bit i <-> key ks0 + i):
        //   keys of earlier timesteps: all; of the query's timestep: every state token (offset % 3 == 0) and the query's
        //   own agent's tokens up to the query itself; later timesteps and keys >= Lk: none.
        asm volatile("" ::: "memory");   // keep this a real (scalar) branch: hipcc otherwise speculates the mask math for every sub-tile
        auto ones = [](int n) -> unsigned { return n >= 32 ? 0xFFFFFFFFu : (n <= 0 ? 0u : ((1u << n) - 1u)); };
        unsigned vis_all, bias_all = 0u;
        if (rep_tile) {
          // representative keys j = 3 t + k: visible while j <= 3 tq (earlier steps, and the state token of the query's step),
          // m-fold; the representative's own queries also see their tokens 1..kq of step tq, once
          const int bias_end = 3 * tq + 1 - j0;
          bias_all = ones(min(bias_end, rep_keys - j0));
          vis_all = ones(min(bias_end + (rep_q ? kq : 0), rep_keys - j0));
        } else {
        const int same0 = tq * A3 - ks0;                                   // first key of the query's timestep
        const unsigned before = ones(same0);
        const unsigned same = ones(min(same0 + A3, Lk - ks0)) & ~before;
        int off3 = (ks_t0 - ks0) % 3;                                      // ks_t0 <= ks0: first state token at or after ks0
        off3 = off3 < 0 ? off3 + 3 : off3;
        const unsigned every3 = (unsigned)(0x249249249249ull << off3);
        // variant 3 (Decision Transformer, token order rtg, state, action — kept in the slots state, rtg, action): a state
        // token also sees its own agent's rtg token, one position AFTER it
        const unsigned own = rep_q ? 0u : (ones(pos - ks0 + 1 + ((variant == 3 && kq == 0) ? 1 : 0)) & ~ones(pos - kq - ks0));
        unsigned before_v = before;
        if (variant == 4) {
          // cfg.model.attend_own_return_action (utils/train_utils.py:114-129; round 6): of the EARLIER timesteps a query sees the state
          // tokens and its own agent's return / action tokens only.  Own tokens of type k sit at offset 3 aq + k of every timestep:
          // bits (3 aq + k - r0) mod A3 + m A3 of this sub-tile, r0 = offset of its first key in its timestep (plain contexts only)
          const int r0 = ks0 - ks_t0;
          unsigned ownpat = 0u;
#pragma unroll
          for (int k = 1; k < 3; ++k) {
            int f = (3 * aq + k - r0) % A3;
            f = f < 0 ? f + A3 : f;
            for (int p = f; p < 32; p += A3) ownpat |= 1u << p;
          }
          before_v &= every3 | ownpat;
        }
        vis_all = before_v | ((every3 | own) & same);
        if (variant) {
          // the 3-slot token layout is kept for the baselines of cfgs/model/{il,trajeglish}.yaml; the token types they do not
          // have are dead as keys.  IL (state, action): rtg keys invisible.  Trajeglish (action only): action keys of earlier
          // steps and of the WHOLE current step (get_causal_mask with one token type: every same-step token is "the state").
          int o2 = off3 + 2;
          o2 = o2 >= 3 ? o2 - 3 : o2;
          const unsigned actions = (unsigned)(0x249249249249ull << o2);
          if (variant == 1) vis_all &= every3 | actions;
          else if (variant == 2) vis_all = (before | same) & actions;
        }
        }
        const unsigned vis = vis_all >> (4 * half);
        if (rep_tile) {
          const unsigned bia = bias_all >> (4 * half);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int pos_r = (r & 3) + 8 * (r >> 2);
            const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)vis, pos_r, 1), mb = (unsigned)__builtin_amdgcn_sbfe((int)bia, pos_r, 1);
            const float v = s0[g][r] + __uint_as_float(__float_as_uint(log2m) & mb);
            sc[r] = __uint_as_float((__float_as_uint(v) & m) | (0xFF800000u & ~m));
          }
        } else {
          // two instructions per score: the key's bit sign-extended to a word mask (v_bfe_i32), then a bit-field insert that keeps the
          // score where the mask is set and -inf elsewhere (v_bfi_b32) — instead of and / compare / select
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)vis, (r & 3) + 8 * (r >> 2), 1);
            sc[r] = __uint_as_float((__float_as_uint(s0[g][r]) & m) | (0xFF800000u & ~m));
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = s0[g][r];
      }
      // ---- online softmax (scores are relative to m_base)
      // Speculative fast path: exponentiate against the CURRENT base and look at the row sums only.  The base has to move when
      // a probability would leave the fp16 range of the split's leading plane; probabilities are non-negative, so
      // psum < 2^15 bounds every one of them (and catches inf / NaN) without computing the maximum.  Softmax does not depend
      // on the base, the accumulators simply keep their scale.  The general path below (exact maximum, rescale) runs while
      // a query has no base yet and in the sub-tiles that fail the test — rare: moving the base at every new maximum sent
      // ~60 % of the sub-tiles of a 32-query wave through it (one of 32 queries sees a new maximum almost every time).
#if ATT_PRIO == 2
      if (!DIR) __builtin_amdgcn_s_setprio(1);       // experiment: the softmax (the VALU chain between the two MFMA chains) at raised priority
#endif
      float psum = 0.f;
      bool general = __any(m_run[g] == NEG_INF);
      if (!general) {
        float pe[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pe[r] = __builtin_amdgcn_exp2f(sc[r]);
          psum += pe[r];
        }
        general = __any(!(psum < ATT_PMAX));
        if (!general) {
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[r] = pe[r];
          l_run[g] += psum;
        }
      }
      if (general) {
        float tmax = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
        {  // the other 16 keys of this query live in lane ^ 32: one v_permlane32_swap instead of an LDS bpermute round trip
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
          tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_rel_old = m_run[g] - m_base[g];                 // 0, or -inf before the first visible key
        const float m_rel_new = fmaxf(m_rel_old, tmax);
        const float shift = (m_rel_new == NEG_INF) ? 0.f : m_rel_new;
        const float alpha = __builtin_amdgcn_exp2f(m_rel_old - shift);
        psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[r] = __builtin_amdgcn_exp2f(sc[r] - shift);
          psum += sc[r];
        }
        l_run[g] = l_run[g] * alpha + psum;                          // per-lane partial (own 16 keys); halves are added at the end
        m_run[g] = (m_rel_new == NEG_INF) ? NEG_INF : m_base[g] + shift;
#pragma unroll
        for (int r = 0; r < 16; ++r) oa[g][r] *= alpha;
      }
      // ---- P^T fragments: k-step kk uses accumulator registers 8*kk .. 8*kk+7 (slot j <-> register 8*kk + j)
      opx8 pf[2][NPL];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        split_frag(sc + 8 * kk, pf[kk]);
      }
      // ---- O^T += V^T . P^T
      if (!V_EARLY) load_v();
#define PV(PA, PB)                                \
  oa[g] = MFMA_OP(v0f[PA], pf[0][PB], oa[g]);     \
  oa[g] = MFMA_OP(v1f[PA], pf[1][PB], oa[g]);
#if ATT_PRIO == 2
      if (!DIR) __builtin_amdgcn_s_setprio(0);
#endif
      PROD_LIST(PV)
#undef PV
      }
    }
    if (RES) continue;
    if (more) sstore(nxt);
    if (PRE && !DIR) {
      // tile it + 1 must have landed; the pieces of tile it + NBUF - 1 requested at the top of this tile (NBUF = 3) stay in flight across
      // the barrier: requests return in order, so a count of this wave's younger pieces is exact
      if (NBUF > 2 && ahead) asm volatile("s_waitcnt vmcnt(%0) ; KV-DMA landed (next tile)" :: "n"(KV_PIECES * 4 / NW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) ; KV-DMA landed" ::: "memory");
    }
    if (!DIR || MODE == MODE6_KEYPAD) __syncthreads();       // DIR: only the key-padding bias block goes through LDS
  }

  // ---- normalise, store rows (ATT_DIRECT_STORE: from the accumulators; else, and for the one-wave streaming form, transposed through LDS)
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    float lr = l_run[g];
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lr), __float_as_uint(lr), false, false);
      lr = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float inv = lr > 0.f ? 1.0f / lr : 0.f;
    if (RES || (ATT_DIRECT_STORE && !DIR && !(ldo & 3))) {
      // straight from the accumulators: lane (query l31, half) holds d = 8 q + 4 half + (0..3) in registers 4 q .. 4 q + 3 — four 16-byte
      // stores into its query's row (a lane pair completes 32 contiguous bytes); the resident tiles leave no LDS for a transpose
      const int gq = qb + (wave * QG + g) * 32 + l31;
      if (gq >= 0 && gq < Lq) {
        float* orow = O + (size_t)b * o_batch_stride + (size_t)gq * ldo + h * HD + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(orow + 8 * q) = f32x4{oa[g][4 * q] * inv, oa[g][4 * q + 1] * inv, oa[g][4 * q + 2] * inv, oa[g][4 * q + 3] * inv};
      }
      continue;
    }
    float* ot = DIR ? reinterpret_cast<float*>(arena + 2 * PADSZ) : reinterpret_cast<float*>(arena) + (wave * QG + g) * (32 * 33);
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[l31 * 33 + mfma_row(r, half)] = oa[g][r] * inv;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = i * 2 + half;
      const int gq = qb + (wave * QG + g) * 32 + q;
      if (gq >= 0 && gq < Lq) O[(size_t)b * o_batch_stride + (size_t)gq * ldo + h * HD + l31] = ot[q * 33 + l31];
    }
  }
  }   // query blocks of this workgroup
  if (ab.cprof) {
    __syncthreads();
    if (tid == 0) {
      const int slot = 2 * min(A + (rep_keys > 0 ? 1 : 0), 31);        // context slots A' (compact classes: regular + representative)
      atomicAdd(ab.cprof + slot, __builtin_amdgcn_s_memtime() - t_start);
      atomicAdd(ab.cprof + slot + 1, 1ull);
    }
  }
}

// ---- K / V rows -> split images (the layout the PRE kernel stages by DMA) -------------------------------------------
// image of (context b, head h, tile kt) at img + ((b*NHEAD + h)*nkt + kt) * KV_IMG:
//   K part  [plane 3][d>>3 4][key 64][d&7 8]          V^T part  [plane 3][key>>2 16][d 32][key&3 4]
// Full mode: one block per (tile, head, context) builds the 24 KB image in LDS and writes it out in 16-byte pieces;
// keys >= Lk are zero (P = 0 times a garbage V would be NaN).
__global__ __launch_bounds__(256) void kv_split_kernel(const float* __restrict__ K, const float* __restrict__ V, int ldkv,
                                                       long kv_batch_stride, int Lk, int nkt, op_t* __restrict__ img) {
  constexpr int K_PLANE = KT6 * HD, V_PLANE = HD * KT6;
  __shared__ __attribute__((aligned(16))) op_t im[KV_IMG];
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, k0 = kt * KT6;
  const float* Kb = K + (size_t)b * kv_batch_stride + h * HD;
  const float* Vb = V + (size_t)b * kv_batch_stride + h * HD;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 pk[2], pv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i, r = idx >> 3, c = (idx & 7) * 4;
    const int kr = k0 + r;
    pk[i] = kr < Lk ? *reinterpret_cast<const f32x4*>(Kb + (size_t)kr * ldkv + c) : zero4;
    const int vr = k0 + 2 * (tid >> 3) + i;
    pv[i] = vr < Lk ? *reinterpret_cast<const f32x4*>(Vb + (size_t)vr * ldkv + (tid & 7) * 4) : zero4;
  }
  op_t* Kd = im;
  op_t* Vd = im + NPL * K_PLANE;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i, r = idx >> 3, c = (idx & 7) * 4;
    u32x2 kp[NPL];
    split_quad(pk[i], kp);
    const int ko = ((c >> 3) * KT6 + r) * 8 + (c & 7);
#pragma unroll
    for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(Kd + q * K_PLANE + ko) = kp[q];
  }
  {
    const int rp = tid >> 3, c = (tid & 7) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned vp[NPL];
      split_pair(pv[0][e], pv[1][e], vp);
      const int vo = ((rp >> 1) * HD + (c + e)) * 4 + (rp & 1) * 2;
#pragma unroll
      for (int q = 0; q < NPL; ++q) *reinterpret_cast<unsigned*>(Vd + q * V_PLANE + vo) = vp[q];
    }
  }
  __syncthreads();
  u32x4* dst = reinterpret_cast<u32x4*>(img + (((size_t)b * NHEAD + h) * nkt + kt) * KV_IMG);
  const u32x4* srcv = reinterpret_cast<const u32x4*>(im);
#pragma unroll
  for (int i = 0; i < KV_PIECES; ++i) dst[tid + 256 * i] = srcv[tid + 256 * i];
}

// Rows mode (KV cache updates): row r of context b (K + b*kv_batch_stride + r*ldkv) goes to key position pos[r] of
// the context's images; one thread per (context, row, 4 consecutive dims).
__global__ __launch_bounds__(256) void kv_split_rows_kernel(const float* __restrict__ K, const float* __restrict__ V,
                                                            int ldkv, long kv_batch_stride, const int* __restrict__ pos,
                                                            int B, int R, int nkt, op_t* __restrict__ img) {
  constexpr int K_PLANE = KT6 * HD, V_PLANE = HD * KT6;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long)B * R * 64) return;
  const int g = (int)(gid & 63), r = (int)((gid >> 6) % R), b = (int)((gid >> 6) / R);
  const int col = g * 4, h = col >> 5, d = col & 31;
  const int p = pos[r], kt = p >> 6, key = p & 63;
  const f32x4 kx = *reinterpret_cast<const f32x4*>(K + (size_t)b * kv_batch_stride + (size_t)r * ldkv + col);
  const f32x4 vx = *reinterpret_cast<const f32x4*>(V + (size_t)b * kv_batch_stride + (size_t)r * ldkv + col);
  op_t* Kd = img + (((size_t)b * NHEAD + h) * nkt + kt) * KV_IMG;
  op_t* Vd = Kd + NPL * K_PLANE;
  u32x2 kp[NPL];
  split_quad(kx, kp);
  const int ko = ((d >> 3) * KT6 + key) * 8 + (d & 7);
#pragma unroll
  for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(Kd + q * K_PLANE + ko) = kp[q];
  unsigned short* Vs = reinterpret_cast<unsigned short*>(Vd);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned vp[NPL];
    split_pair(vx[e], 0.f, vp);
    const int vo = ((key >> 2) * HD + (d + e)) * 4 + (key & 3);
#pragma unroll
    for (int q = 0; q < NPL; ++q) Vs[q * V_PLANE + vo] = (unsigned short)vp[q];
  }
}

// Zero the keys >= k0 of tile `tile` of every (context, head): producers that write images row by row (the fused QKV GEMM
// epilogue) leave the tail of the last tile of a key region untouched, and the attention kernel stages whole tiles.
__global__ __launch_bounds__(256) void kv_zero_tail_kernel(int k0, int tile, int nkt, op_t* __restrict__ img) {
  constexpr int K_PLANE = KT6 * HD, V_PLANE = HD * KT6;
  op_t* base = img + ((size_t)blockIdx.x * nkt + tile) * KV_IMG;   // k0 = first invalid key of the tile (multiple of 4)
  const int nk = KT6 - k0;
  for (int i = threadIdx.x; i < 4 * NPL * nk * 4; i += 256) {     // K: 4 NPL (plane, d>>3) runs of nk keys x 8 elements = nk*4 dwords
    const int run = i / (nk * 4), off = i - run * (nk * 4);
    reinterpret_cast<unsigned*>(base + run * KT6 * 8 + k0 * 8)[off] = 0u;
  }
  const int q0 = k0 >> 2, nq = 16 - q0;
  for (int i = threadIdx.x; i < NPL * nq * 64; i += 256) {     // V^T: NPL planes, quads >= q0, 32 d x 4 keys = 64 dwords each
    const int pl = i / (nq * 64), off = i - pl * (nq * 64);
    reinterpret_cast<unsigned*>(base + NPL * K_PLANE + pl * V_PLANE + q0 * HD * 4)[off] = 0u;
  }
}
// keys [key0 + n, end of that tile) of every (context, head): the region of n keys starting at key0 (a multiple of 64)
int launch_kv_zero_tail(int B, int key0, int n, int nkt, void* img, hipStream_t st) {
  if (B <= 0 || n % KT6 == 0) return CTRLSIM_OK;
  const int tile = (key0 + n) / KT6;
  if (!img || (n & 3) || (key0 & 63) || tile >= nkt) return CTRLSIM_EINVAL;
  hipLaunchKernelGGL(kv_zero_tail_kernel, dim3(B * NHEAD), dim3(256), 0, st, n % KT6, tile, nkt, static_cast<op_t*>(img));
  return ctrlsim_launch_status();
}
// The same for up to 2 * MAXC (class, key region) entries in ONE launch: entry e covers the images from tile tile0 on of B contexts
// Round 6: (i) the same entries applied to up to 8 image SETS of equal geometry in one launch (blockIdx.y: the scene encoder's images and the
// memory K / V images of every decoder layer share the classes' tile offsets) — ten launches per forward pass became one; (ii) a tail that
// is a whole 32-key sub-tile is not zeroed at all: no kernel ever computes a sub-tile without a valid key (mask-table code 0, `ks0 >= k_end`,
// `j0 >= rep_need`), so only regions whose length is not a multiple of 32 have a tail anybody reads.
struct KvTailEntry { int wg0, nkt, tile, k0; long tile0; };
struct KvTailBatch { int n; KvTailEntry e[2 * MAXC]; };
struct KvImgSets { op_t* img[8]; };
__global__ __launch_bounds__(256) void kv_zero_tails_kernel(KvTailBatch tb, KvImgSets sets) {
  op_t* __restrict__ img = sets.img[blockIdx.y];
  constexpr int K_PLANE = KT6 * HD, V_PLANE = HD * KT6;
  int ei = 0;
  while (ei + 1 < tb.n && (int)blockIdx.x >= tb.e[ei + 1].wg0) ++ei;
  const KvTailEntry& e = tb.e[ei];
  const int k0 = e.k0;
  const int nk = KT6 - k0, q0 = k0 >> 2, nq = 16 - q0;
  // one workgroup per CONTEXT (all eight heads: 41 000 workgroups of a few hundred stores each had made the launch dispatch-bound)
  for (int h = 0; h < NHEAD; ++h) {
    op_t* base = img + ((size_t)e.tile0 + ((size_t)((int)blockIdx.x - e.wg0) * NHEAD + h) * e.nkt + e.tile) * KV_IMG;
    for (int i = threadIdx.x; i < 4 * NPL * nk * 4; i += 256) {
      const int run = i / (nk * 4), off = i - run * (nk * 4);
      reinterpret_cast<unsigned*>(base + run * KT6 * 8 + k0 * 8)[off] = 0u;
    }
    for (int i = threadIdx.x; i < NPL * nq * 64; i += 256) {
      const int pl = i / (nq * 64), off = i - pl * (nq * 64);
      reinterpret_cast<unsigned*>(base + NPL * K_PLANE + pl * V_PLANE + q0 * HD * 4)[off] = 0u;
    }
  }
}
int launch_kv_zero_tails(int n, const KvTailHost* t, int nimg, void* const* imgs, hipStream_t st) {
  if (n < 0 || n > 2 * MAXC || nimg < 1 || nimg > 8 || !imgs) return CTRLSIM_EINVAL;
  KvImgSets sets;
  for (int i = 0; i < 8; ++i) {
    sets.img[i] = static_cast<op_t*>(imgs[i < nimg ? i : 0]);
    if (!sets.img[i]) return CTRLSIM_EINVAL;
  }
  KvTailBatch tb;
  tb.n = 0;
  int wg = 0;
  for (int i = 0; i < n; ++i) {
    if (t[i].B <= 0 || t[i].n % 32 == 0) continue;            // (a tail of whole sub-tiles is never read: see the kernel)
    const int tile = (t[i].key0 + t[i].n) / KT6;
    if ((t[i].n & 3) || (t[i].key0 & 63) || tile >= t[i].nkt) return CTRLSIM_EINVAL;
    tb.e[tb.n++] = KvTailEntry{wg, t[i].nkt, tile, t[i].n % KT6, t[i].tile0};
    wg += t[i].B;
  }
  if (tb.n == 0) return CTRLSIM_OK;
  hipLaunchKernelGGL(kv_zero_tails_kernel, dim3(wg, nimg), dim3(256), 0, st, tb, sets);
  return ctrlsim_launch_status();
}

// Rows mode for up to MAXC classes in one launch: class c holds B contexts of R rows starting at row row0 of K / V (row stride ldkv,
// context stride R * ldkv), written at key pos[r] of the images from tile tile0 on (nkt tiles per (context, head)).
struct KvRowsClass { long g0, row0, tile0; const int* pos; int R, nkt; };
struct KvRowsBatch { int n; KvRowsClass c[MAXC]; };
__global__ __launch_bounds__(256) void kv_split_rows_classes_kernel(const float* __restrict__ K, const float* __restrict__ V, int ldkv,
                                                                    KvRowsBatch kb, long total, op_t* __restrict__ img) {
  constexpr int K_PLANE = KT6 * HD, V_PLANE = HD * KT6;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  int ci = 0;
  while (ci + 1 < kb.n && gid >= kb.c[ci + 1].g0) ++ci;
  const KvRowsClass& c = kb.c[ci];
  const long l = gid - c.g0;
  const int g = (int)(l & 63), r = (int)((l >> 6) % c.R), b = (int)((l >> 6) / c.R);
  const int col = g * 4, h = col >> 5, d = col & 31;
  const int p = c.pos[r], kt = p >> 6, key = p & 63;
  const size_t row = (size_t)c.row0 + (size_t)b * c.R + r;
  const f32x4 kx = *reinterpret_cast<const f32x4*>(K + row * ldkv + col);
  const f32x4 vx = *reinterpret_cast<const f32x4*>(V + row * ldkv + col);
  op_t* Kd = img + ((size_t)c.tile0 + ((size_t)b * NHEAD + h) * c.nkt + kt) * KV_IMG;
  op_t* Vd = Kd + NPL * K_PLANE;
  u32x2 kp[NPL];
  split_quad(kx, kp);
  const int ko = ((d >> 3) * KT6 + key) * 8 + (d & 7);
#pragma unroll
  for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2*>(Kd + q * K_PLANE + ko) = kp[q];
  unsigned short* Vs = reinterpret_cast<unsigned short*>(Vd);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned vp[NPL];
    split_pair(vx[e], 0.f, vp);
    const int vo = ((key >> 2) * HD + (d + e)) * 4 + (key & 3);
#pragma unroll
    for (int q = 0; q < NPL; ++q) Vs[q * V_PLANE + vo] = (unsigned short)vp[q];
  }
}
int launch_kv_split_rows_classes(const float* K, const float* V, int ldkv, int n, const KvRowsHost* cls, void* img, hipStream_t st) {
  if (n < 0 || n > MAXC || !K || !V || !img || (ldkv & 3)) return CTRLSIM_EINVAL;
  KvRowsBatch kb;
  kb.n = 0;
  long total = 0;
  for (int i = 0; i < n; ++i) {
    if (cls[i].B <= 0 || cls[i].R <= 0) continue;
    if (!cls[i].pos) return CTRLSIM_EINVAL;
    kb.c[kb.n++] = KvRowsClass{total, cls[i].row0, cls[i].tile0, cls[i].pos, cls[i].R, cls[i].nkt};
    total += (long)cls[i].B * cls[i].R * 64;
  }
  if (kb.n == 0) return CTRLSIM_OK;
  hipLaunchKernelGGL(kv_split_rows_classes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, K, V, ldkv, kb, total,
                     static_cast<op_t*>(img));
  return ctrlsim_launch_status();
}

int launch_kv_split(const float* K, const float* V, int ldkv, long kv_batch_stride, int B, int Lk, int nkt, void* img,
                    hipStream_t st) {
  if (B <= 0 || Lk <= 0) return CTRLSIM_OK;
  if (!K || !V || !img || (ldkv & 3) || nkt * KT6 < Lk) return CTRLSIM_EINVAL;
  hipLaunchKernelGGL(kv_split_kernel, dim3(nkt, NHEAD, B), dim3(256), 0, st, K, V, ldkv, kv_batch_stride, Lk, nkt,
                     static_cast<op_t*>(img));
  return ctrlsim_launch_status();
}
int launch_kv_split_rows(const float* K, const float* V, int ldkv, long kv_batch_stride, const int* pos, int B, int R,
                         int nkt, void* img, hipStream_t st) {
  if (B <= 0 || R <= 0) return CTRLSIM_OK;
  if (!K || !V || !img || !pos || (ldkv & 3)) return CTRLSIM_EINVAL;
  const long total = (long)B * R * 64;
  hipLaunchKernelGGL(kv_split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, K, V, ldkv,
                     kv_batch_stride, pos, B, R, nkt, static_cast<op_t*>(img));
  return ctrlsim_launch_status();
}

static double attn_pairs(int mode, const int* q_pos, int Lq, int Lk, int A, int rep_keys = 0) {
  if (mode == MODE6_CAUSAL && !q_pos) {
    const double A3 = 3.0 * A, T = (double)Lk / A3;
    double pairs = A3 * A3 * T * (T - 1) / 2.0 + T * A * (3.0 * A + 3.0);
    if (rep_keys > 0)   // regular queries x representative keys (3t + 1 each), representative queries x (regular + own) keys
      pairs += A3 * (3.0 * T * (T - 1) / 2.0 + T) + 3.0 * A3 * T * (T - 1) / 2.0 + 3.0 * A * T + 9.0 * T * (T - 1) / 2.0 + 6.0 * T;
    return pairs;
  }
  return (double)Lq * (double)(Lk + rep_keys);
}

int launch_attention_bf16x6(int mode, const float* Q, int ldq, long q_batch_stride, const float* K, const float* V,
                            int ldkv, long kv_batch_stride, float* O, int ldo, long o_batch_stride, const int* q_pos,
                            const unsigned char* key_pad, int B, int Lq, int Lk, int A, hipStream_t st) {
  if (B <= 0 || Lq <= 0) return CTRLSIM_OK;
  if (Lk <= 0 || (ldq & 3) || (ldkv & 3)) return CTRLSIM_EINVAL;
  if (mode < 0 || mode > 5 || (mode == MODE6_KEYPAD && !key_pad)) return CTRLSIM_EINVAL;
  const int variant = mode >= MODE6_CAUSAL ? mode - MODE6_CAUSAL : 0;   // mode 1 CtRL-Sim mask, 2 IL, 3 Trajeglish, 4 DT, 5 CtRL-Sim with attend_own_return_action
  mode = mode >= MODE6_CAUSAL ? MODE6_CAUSAL : MODE6_KEYPAD;
  AttnBatch ab;
  ab.n = 1;
  ab.cprof = nullptr;
  ab.c[0] = AttnClass{0, 0, 0, 0, q_batch_stride, o_batch_stride, kv_batch_stride, q_pos, nullptr, Lq, Lk, A, 0, Lk, (Lq + 127) / 128, 0, 0.f};
  dim3 g(ab.c[0].qblocks * NHEAD * B), blk(256);
  const float scale = 0.17677669529663687f * 1.4426950408889634f;  // log2(e)/sqrt(32)
  prof_before(PROF_ATTN, st);
  if (mode == MODE6_CAUSAL) {
    hipLaunchKernelGGL((attention_bf16x6_kernel<MODE6_CAUSAL, false>), g, blk, 0, st, Q, ldq, K, V, ldkv, O, ldo, key_pad, scale,
                       variant, ab);
  } else {
    hipLaunchKernelGGL((attention_bf16x6_kernel<MODE6_KEYPAD, false>), g, blk, 0, st, Q, ldq, K, V, ldkv, O, ldo, key_pad, scale, 0,
                       ab);
  }
  prof_after(PROF_ATTN, attn_pairs(mode, q_pos, Lq, Lk, A) * 128.0 * NHEAD * B, st,
             (double)B * (8.0 * DM * Lq + 8.0 * DM * Lk),
             mode == MODE6_CAUSAL ? PKIND_ATTN_CAUSAL : PKIND_ATTN_KEYPAD);
  return ctrlsim_launch_status();
}

// K / V from split images (launch_kv_split*), n classes of contexts in one launch (n <= MAXC).  Per class: nkt tiles per (context,
// head) from tile img_tile0 of `img` on.  rep_keys > 0 (causal mask of the CtRL-Sim model only): compact contexts — Lk regular
// keys in tiles [0, ceil(Lk / 64)), and rep_keys representative keys of multiplicity rep_mult in the tiles from
// ceil(rep_pos0 / 64) on (see the kernel header); rep_pos0 >= Lk is the regular length of the full window (the K/V cache
// layout); query positions (row index, or q_pos) >= rep_pos0 address the representative's own tokens.
int launch_attention_classes(int mode, const float* Q, int ldq, const void* img, float* O, int ldo, const unsigned char* key_pad,
                             int n, const AttnClassHost* cls, hipStream_t st) {
  if (n < 1 || n > MAXC || !cls || (ldq & 3) || !img) return CTRLSIM_EINVAL;
  if (mode < 0 || mode > 5 || (mode == MODE6_KEYPAD && !key_pad)) return CTRLSIM_EINVAL;
  const int variant = mode >= MODE6_CAUSAL ? mode - MODE6_CAUSAL : 0;
  mode = mode >= MODE6_CAUSAL ? MODE6_CAUSAL : MODE6_KEYPAD;
  AttnBatch ab;
  ab.n = 0;
  ab.cprof = (mode == MODE6_CAUSAL && cls[0].q_pos == nullptr) ? ctrlsim_attn_cprof_ptr() : nullptr;   // the launches over the token rows
  int wg = 0;
  double flops = 0.0, bytes = 0.0;
  bool use_tbl = mode == MODE6_CAUSAL && variant == 0 && ctrlsim_option(OPT_ATTN_TBL) != 0;
  // few-query launches (at most three 32-query groups per context and head: the second pass, the last layer on the queried rows, the
  // K/V-cached steps): the streaming form of the kernel, one wave per workgroup
  // (option value 2, an experiment: EVERY launch in the streaming form — each 32-query wave then re-reads its K / V range from L2)
  bool dir = ctrlsim_option(OPT_ATTN_DIRECT) != 0;
  const bool dir_all = ctrlsim_option(OPT_ATTN_DIRECT) == 2;
  for (int k = 0; k < n; ++k) {
    const AttnClassHost& c = cls[k];
    if (c.B <= 0 || c.Lq <= 0) continue;
    if (c.Lq > 96 && !dir_all) dir = false;
    // mask-table kernel: every class brings its table, the query rows are the token rows, CtRL-Sim mask, keys = the whole row layout
    use_tbl = use_tbl && c.mask_tbl && !c.q_pos && (c.rep_keys == 0 || c.rep_pos0 == c.Lk);
  }
  // the two full-row kernels run ATT_NW_FULL waves of ATT_QG_FULL 32-query groups each (256 queries per workgroup), the
  // in-kernel-mask causal kernel 4 waves of one group
  const int nw = (mode == MODE6_KEYPAD || use_tbl) ? ATT_NW_FULL : 4, qg = (mode == MODE6_KEYPAD || use_tbl) ? ATT_QG_FULL : 1;
  // key-padded launches whose keys fit four tiles per (context, head) (the scene: 200 polylines + the vehicles): the resident-keys kernel
#ifndef ATT_KEYPAD_RES
#define ATT_KEYPAD_RES 1
#endif
  // (two-plane scheme only: four three-plane stages are 100 KB — one workgroup per CU)
  bool res = ATT_KEYPAD_RES && NPL == 2 && mode == MODE6_KEYPAD && !dir && ATT_QG_FULL == 1 && !(ldo & 3);
  long pairs_total = 0;
  for (int k = 0; k < n; ++k) {
    if (cls[k].B <= 0 || cls[k].Lq <= 0) continue;
    res = res && (cls[k].Lk + KT6 - 1) / KT6 <= 4 && !cls[k].q_pos;
    pairs_total += (long)cls[k].B * NHEAD;
  }
  for (int k = 0; k < n; ++k) {
    AttnClassHost c = cls[k];
    if (c.B <= 0 || c.Lq <= 0) continue;
    if (c.rep_keys == 0) c.rep_pos0 = c.Lk;
    if (c.Lk <= 0 || c.rep_keys < 0 || c.rep_pos0 < c.Lk ||
        c.nkt < (c.rep_pos0 + KT6 - 1) / KT6 + (c.rep_keys + KT6 - 1) / KT6)
      return CTRLSIM_EINVAL;
    if (c.rep_keys > 0 && (mode != MODE6_CAUSAL || variant || c.rep_mult < 1 || c.Lk % (3 * c.A) || c.rep_pos0 % (3 * c.A)))
      return CTRLSIM_EINVAL;
    int qblocks = dir ? (c.Lq + 31) / 32 : (c.Lq + 32 * nw * qg - 1) / (32 * nw * qg);
    if (res) {
      // workgroups per (context, head): one walks all query blocks when the launch has enough (context, head) pairs to fill the chip
      // (two workgroups per CU: 512 resident), else the blocks are dealt to a few
      const int want = (int)((1536 + pairs_total - 1) / pairs_total);
      qblocks = want < 1 ? 1 : (want > qblocks ? qblocks : want);
    }
    ab.c[ab.n++] = AttnClass{c.q_row0 * ldq, c.o_row0 * ldo, c.img_tile0, c.pad_off, c.q_bs, c.o_bs, (long)c.nkt, c.q_pos,
                             static_cast<const unsigned long long*>(c.mask_tbl), c.Lq, c.Lk, c.A, c.rep_keys, c.rep_pos0, qblocks, wg,
                             c.rep_keys > 0 ? log2f((float)c.rep_mult) : 0.f};
    wg += qblocks * NHEAD * c.B;
    flops += attn_pairs(mode, c.q_pos, c.Lq, c.Lk, c.A, c.rep_keys) * 128.0 * NHEAD * c.B;
    bytes += (double)c.B * (8.0 * DM * c.Lq + 4.0 * NPL * DM * (c.Lk + c.rep_keys));   // Q in + O out (fp32), K and V images (NPL planes)
  }
  if (ab.n == 0) return CTRLSIM_OK;
  dim3 g(wg), blk(dir ? 64 : 64 * nw);
  const float scale = 0.17677669529663687f * 1.4426950408889634f;
  const float* imgf = static_cast<const float*>(img);
  prof_before(PROF_ATTN, st);
  if (dir && mode == MODE6_CAUSAL && use_tbl) {
    hipLaunchKernelGGL((attention_bf16x6_kernel<MODE6_CAUSAL, true, true, true>), g, blk, 0, st, Q, ldq, imgf, nullptr, 0, O, ldo, key_pad,
                       scale, variant, ab);
  } else if (dir && mode == MODE6_CAUSAL) {
    hipLaunchKernelGGL((attention_bf16x6_kernel<MODE6_CAUSAL, true, false, true>), g, blk, 0, st, Q, ldq, imgf, nullptr, 0, O, ldo, key_pad,
                       scale, variant, ab);
  } else if (dir) {
    hipLaunchKernelGGL((attention_bf16x6_kernel<MODE6_KEYPAD, true, false, true>), g, blk, 0, st, Q, ldq, imgf, nullptr, 0, O, ldo, key_pad,
                       scale, 0, ab);
  } else if (mode == MODE6_CAUSAL && use_tbl) {
    hipLaunchKernelGGL((attention_bf16x6_kernel<MODE6_CAUSAL, true, true, false, ATT_NW_FULL, ATT_QG_FULL>), g, blk, 0, st, Q, ldq, imgf, nullptr, 0, O, ldo, key_pad, scale,
                       variant, ab);
  } else if (mode == MODE6_CAUSAL) {
    hipLaunchKernelGGL((attention_bf16x6_kernel<MODE6_CAUSAL, true>), g, blk, 0, st, Q, ldq, imgf, nullptr, 0, O, ldo, key_pad, scale,
                       variant, ab);
  } else if (res) {
    if constexpr (NPL == 2)
      hipLaunchKernelGGL((attention_bf16x6_kernel<MODE6_KEYPAD, true, false, false, 8, 1, true>), g, blk, 0, st, Q, ldq, imgf, nullptr, 0, O, ldo, key_pad, scale, 0,
                         ab);
  } else {
    hipLaunchKernelGGL((attention_bf16x6_kernel<MODE6_KEYPAD, true, false, false, ATT_NW_FULL, ATT_QG_FULL>), g, blk, 0, st, Q, ldq, imgf, nullptr, 0, O, ldo, key_pad, scale, 0,
                       ab);
  }
  prof_after(PROF_ATTN, flops, st, bytes, mode == MODE6_CAUSAL ? PKIND_ATTN_CAUSAL : PKIND_ATTN_KEYPAD);
  return ctrlsim_launch_status();
}

// Mask tables of n classes (cls[k].mask_tbl: attn_mask_table_bytes(Lq, nkt) bytes each; classes without a table are skipped), one launch
size_t attn_mask_table_bytes(int Lq, int nkt) {
  const size_t groups = (size_t)4 * ((Lq + 127) / 128);
  return groups * 2 * nkt * TBL_ENTRY * sizeof(u64) + ((groups * TBL_CTL(nkt) * sizeof(u32) + 255) & ~size_t(255));
}
int launch_attn_mask_tables(int n, const AttnClassHost* cls, hipStream_t st) {
  if (n < 0 || n > MAXC || !cls) return CTRLSIM_EINVAL;
  TblBatch tb;
  tb.n = 0;
  int wg = 0;
  for (int k = 0; k < n; ++k) {
    const AttnClassHost& c = cls[k];
    if (c.B <= 0 || c.Lq <= 0 || !c.mask_tbl) continue;
    const int rp0 = c.rep_keys ? c.rep_pos0 : c.Lk;
    if (c.Lk <= 0 || c.A <= 0 || c.nkt < (rp0 + KT6 - 1) / KT6 + (c.rep_keys + KT6 - 1) / KT6 || c.nkt > 128) return CTRLSIM_EINVAL;
    const int groups = 4 * ((c.Lq + 127) / 128);
    tb.c[tb.n++] = TblClass{static_cast<u64*>(const_cast<void*>(c.mask_tbl)), c.Lq, c.Lk, c.A, c.rep_keys, rp0, c.nkt, groups, wg};
    wg += groups;
  }
  if (tb.n == 0) return CTRLSIM_OK;
  hipLaunchKernelGGL(attn_mask_table_kernel, dim3(wg), dim3(256), 0, st, tb);
  return ctrlsim_launch_status();
}

int launch_attention_bf16x6_pre(int mode, const float* Q, int ldq, long q_batch_stride, const void* img, int nkt, float* O,
                                int ldo, long o_batch_stride, const int* q_pos, const unsigned char* key_pad, int B, int Lq,
                                int Lk, int A, int rep_keys, int rep_mult, int rep_pos0, const void* mask_tbl, hipStream_t st) {
  if (B <= 0 || Lq <= 0) return CTRLSIM_OK;
  const AttnClassHost c{B, Lq, Lk, A, rep_keys, rep_mult, rep_pos0, nkt, 0, q_batch_stride, 0, o_batch_stride, 0, 0, q_pos, mask_tbl};
  return launch_attention_classes(mode, Q, ldq, img, O, ldo, key_pad, 1, &c, st);
}

}  // namespace SPLIT_NS
