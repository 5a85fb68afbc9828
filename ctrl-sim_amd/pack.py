"""Weight packing for the HIP forward: the reference state_dict tensors plus the "fold.*" tensors.

Folds (all in float64, rounded once to float32) are products of weights only — no data dependence — and leave
the model function unchanged in exact arithmetic:

  fold.embed_state.w   = W_sg[:, :D] @ W_state3            encoder.embed_state.mlp.3 then the state half of
  fold.embed_goal.w    = W_sg[:, D:] @ W_goal3             encoder.embed_state_goal (modules/encoder.py:21-23,106)
  fold.embed_goal.b    = W_sg[:, :D] b_state3 + W_sg[:, D:] b_goal3 + b_sg
  fold.rtg_table_{goal,veh,road} = E_c @ W_rtg[:, cD:(c+1)D]^T   embed_rtg over three embedding rows (:116-125)
  fold.map.U, fold.map.cb, fold.map.Mt, fold.map.mb         single-seed attention pooling of the map encoder
  fold.map.Wc, fold.map.Wc2, fold.map.G                                   first point-MLP layer + its LayerNorm in closed form
                                                             (modules/map_encoder.py:44-46; see csrc/map_encoder.hip)
Tensors are laid out back to back in ONE float32 buffer, each aligned to 256 bytes; `names`/`offsets` (in floats)
are handed to ctrlsim_model_create.
"""
from __future__ import annotations

import numpy as np

from .spec import Dims


def fold(dims: Dims, w: dict) -> dict:
    D, H = dims.D, dims.H
    dh = D // H
    f8 = lambda k: np.asarray(w[k], np.float64)
    out = {}
    Wsg, bsg = f8("encoder.embed_state_goal.weight"), f8("encoder.embed_state_goal.bias")
    Ws3, bs3 = f8("encoder.embed_state.mlp.3.weight"), f8("encoder.embed_state.mlp.3.bias")
    Wg3, bg3 = f8("encoder.embed_goal.mlp.3.weight"), f8("encoder.embed_goal.mlp.3.bias")
    out["fold.embed_state.w"] = Wsg[:, :D] @ Ws3
    out["fold.embed_goal.w"] = Wsg[:, D:] @ Wg3
    out["fold.embed_goal.b"] = Wsg[:, :D] @ bs3 + Wsg[:, D:] @ bg3 + bsg
    Wr = f8("encoder.embed_rtg.weight")
    rtg_bias = f8("encoder.embed_rtg.bias").copy()
    for c, nm in enumerate(("goal", "veh", "road")):
        if getattr(dims, "VARIANT", 0) == 3:
            # decision transformer: embed_rtg(cat_c(w_c r_c + b_c)) = sum_c r_c (W_c w_c) + (sum_c W_c b_c + bias): one row per
            # component instead of a table, the constants folded into the bias
            Wc = Wr[:, c * D:(c + 1) * D]
            out[f"fold.rtg_table_{nm}"] = (Wc @ f8(f"encoder.embed_rtg_{nm}.weight")[:, 0])[None, :]
            rtg_bias += Wc @ f8(f"encoder.embed_rtg_{nm}.bias")
        else:
            out[f"fold.rtg_table_{nm}"] = f8(f"encoder.embed_rtg_{nm}.weight") @ Wr[:, c * D:(c + 1) * D].T
    out["fold.rtg_bias"] = rtg_bias
    pre = "encoder.map_encoder."
    W2, b2 = f8(pre + "road_pts_encoder.mlp.3.weight"), f8(pre + "road_pts_encoder.mlp.3.bias")
    Wi, bi = f8(pre + "road_pts_attn_layer.in_proj_weight"), f8(pre + "road_pts_attn_layer.in_proj_bias")
    seed = f8(pre + "map_seeds").reshape(D)
    q = Wi[:D] @ seed + bi[:D]
    Wk, bk, Wv, bv = Wi[D:2 * D], bi[D:2 * D], Wi[2 * D:], bi[2 * D:]
    U = np.zeros((D, H))
    cb = np.zeros(H)
    M = np.zeros((D, D))
    mb = np.zeros(D)
    sc = 1.0 / np.sqrt(dh)
    for h in range(H):
        sl = slice(h * dh, (h + 1) * dh)
        U[:, h] = (W2.T @ (Wk[sl].T @ q[sl])) * sc
        cb[h] = (q[sl] @ (Wk[sl] @ b2 + bk[sl])) * sc
        M[sl, :] = Wv[sl] @ W2
        mb[sl] = Wv[sl] @ b2 + bv[sl]
    # first point-MLP layer + LayerNorm in closed form: y_c - mean_c(y) = wt_c . (x, y, e, 1) with the column-centred weights
    # wt_c, so  var = (x, y, e, 1)^T G (x, y, e, 1)  (G = sum_c wt_c wt_c^T / D: 10 numbers) and
    # LN(y)_c = (g_c wt_c) . (x, y, e, 1) * rstd + beta_c — the kernel evaluates the layer once per point instead of three times
    W1, b1 = f8(pre + "road_pts_encoder.mlp.0.weight"), f8(pre + "road_pts_encoder.mlp.0.bias")
    g1 = f8(pre + "road_pts_encoder.mlp.1.weight")
    Wt = np.concatenate([W1 - W1.mean(0, keepdims=True), (b1 - b1.mean())[:, None]], axis=1)      # [D, 4]
    G = Wt.T @ Wt / D
    out["fold.map.Wc"] = Wt * g1[:, None]
    # the same with the channels of a pair interleaved per component, [D / 2, 4, 2]: the packed-fp32 kernel reads the pair as one scalar register pair
    out["fold.map.Wc2"] = (Wt * g1[:, None]).reshape(D // 2, 2, 4).transpose(0, 2, 1).copy()
    out["fold.map.G"] = G[np.triu_indices(4)]                            # 00 01 02 03 11 12 13 22 23 33
    out["fold.map.U"] = U
    out["fold.map.cb"] = cb
    out["fold.map.Mt"] = M.T.copy()
    out["fold.map.mb"] = mb
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()}


def bf16_rne(x: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 bits (uint16), round-to-nearest-even (what v_cvt_pk_bf16_f32 does)."""
    b = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = (b + 0x7FFF + ((b >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def bf16_to_f32(h: np.ndarray) -> np.ndarray:
    return (h.astype(np.uint32) << 16).view(np.float32)


def split_scheme():
    """(planes, weight scale) of the operand split compiled into the library (csrc/split.h): (2, 256.0) = two fp16 planes, three
    products; (3, 1.0) = three bf16 planes, six products."""
    from . import _lib
    return (2, 256.0) if int(_lib.lib().ctrlsim_split_scheme()) == 1 else (3, 1.0)


SCHEMES = {1: (2, 256.0), 0: (3, 1.0)}        # ctrlsim_set_option(OPT_SPLIT): 1 = two fp16 planes (x 2^8), 0 = three bf16 planes


def _planes3(W: np.ndarray, scheme=None):
    """[rows, cols] float32 -> the operand planes [NPL][rows][cols] as 16-bit words: W * scale = sum of the planes.
    scheme None = the one currently selected in the library."""
    npl, scale = split_scheme() if scheme is None else SCHEMES[scheme]
    W = np.ascontiguousarray(W, np.float32) * np.float32(scale)
    if npl == 2:
        with np.errstate(over="ignore", invalid="ignore"):
            hi = W.astype(np.float16)                                    # round-to-nearest-even, like v_cvt_pk_f16_f32
            lo = (W - hi.astype(np.float32)).astype(np.float16)
        if not (np.isfinite(hi).all() and np.isfinite(lo).all()):
            raise FloatingPointError(f"weight magnitude {float(np.abs(W).max()) / scale:.4g} is beyond the fp16 range of the "
                                     f"two-plane operand split (|w| < {65504.0 / scale:.0f}): select the three-bf16-plane scheme "
                                     "(ctrlsim_set_option(OPT_SPLIT = 4, 0); RolloutEngine(split='bf16x6'))")
        return np.stack([hi.view(np.uint16), lo.view(np.uint16)], 0)
    hi = bf16_rne(W)
    r1 = W - bf16_to_f32(hi)
    mid = bf16_rne(r1)
    lo = bf16_rne(r1 - bf16_to_f32(mid))
    return np.stack([hi, mid, lo], 0)                                     # [NPL][rows][cols] uint16


def split3_planes(W: np.ndarray, scheme=None) -> np.ndarray:
    """[N,K] float32 -> slab-major operand planes [K/16][NPL][2][N][8] (uint16), each 16-wide k-step stored as two 8-wide half
    planes — the operand layout of csrc/gemm_bf16x6.hip."""
    W = np.ascontiguousarray(W, np.float32)
    N, K = W.shape
    assert K % 16 == 0
    pl = _planes3(W, scheme)
    planes = pl.reshape(pl.shape[0], N, K // 16, 2, 8)                    # [NPL][N][K/16][2][8]
    return np.ascontiguousarray(planes.transpose(2, 0, 3, 1, 4))         # [K/16][NPL][2][N][8]


PLANES_SUFFIX = {1: "#pl1", 0: "#pl0"}


def row_blocks(W: np.ndarray, scheme=None) -> np.ndarray:
    """[N, K] float32 -> operand blocks of 32 output columns [N/32][p][ks K/16][half 2][col 32][8] (uint16) = plane_p(W)[32 cb + col,
    16 ks + 8 half + e]: what the row-stationary in_proj kernel (csrc/gemm_bf16x6.hip: inproj_rs_kernel) streams through LDS, and the
    layout of the fused FFN's W1 blocks."""
    N, K = W.shape
    assert N % 32 == 0 and K % 16 == 0
    pl = _planes3(W, scheme)
    p1 = pl.reshape(pl.shape[0], N // 32, 32, K // 16, 2, 8)              # [p][cb][col][ks][half][e]
    return np.ascontiguousarray(p1.transpose(1, 0, 3, 4, 2, 5))           # [cb][p][ks][half][col][e]


def ffn_planes(W1: np.ndarray, W2: np.ndarray, scheme=None):
    """Operand images of the fused FFN kernel (csrc/ffn_fused.hip), one 48 KB block per 32 hidden units hb:
      W1p[hb][p 3][ks K/16][half 2][row 32][8]   = plane_p(W1)[32 hb + row, 16 ks + 8 half + e]
      W2p[hb][p 3][kk 2][half 2][o D][8]         = plane_p(W2)[o, 32 hb + 16 kk + (j & 3) + 8 (j >> 2) + 4 half]
    (the k-slot order of W2p is the accumulator-register order of the hidden tile, so the ReLU output feeds the second
    product straight from registers).  W1 [F,K], W2 [D,F] float32 -> two uint16 arrays."""
    F, K = W1.shape
    D = W2.shape[0]
    assert W2.shape[1] == F and F % 32 == 0 and K % 16 == 0
    pl1 = _planes3(W1, scheme)
    p1 = pl1.reshape(pl1.shape[0], F // 32, 32, K // 16, 2, 8)            # [p][hb][row][ks][half][e]
    w1p = np.ascontiguousarray(p1.transpose(1, 0, 3, 4, 2, 5))            # [hb][p][ks][half][row][e]
    j = np.arange(8)
    idx = np.empty((2, 2, 8), np.int64)                                   # [kk][half][j] -> hidden offset in the block
    for kk in range(2):
        for half in range(2):
            idx[kk, half] = 16 * kk + (j & 3) + 8 * (j >> 2) + 4 * half
    pl2 = _planes3(W2, scheme)
    p2 = pl2.reshape(pl2.shape[0], D, F // 32, 32)                        # [p][o][hb][hid]
    p2 = p2[:, :, :, idx]                                                 # [p][o][hb][kk][half][j]
    w2p = np.ascontiguousarray(p2.transpose(2, 0, 3, 4, 1, 5))            # [hb][p][kk][half][o][j]
    return w1p, w2p


def ffn_planes_pre(Wo: np.ndarray, W1: np.ndarray, W2: np.ndarray, scheme=1):
    """Operand images of the fused FFN kernel with the out-projection + LayerNorm in front of it as its leading product
    (csrc/ffn_fused.hip, PRE): -> (Wop, W1q, W2p).  Wop = row_blocks(Wo): eight W1-shaped blocks of 32 output columns.  W1q = the W1
    blocks with bits 2 and 3 of k swapped inside every 16-wide k-step — the kernel's first B operand is then the LayerNorm output in
    ACCUMULATOR-register order: slot (half, e) of k-step ks holds k = 16 ks + 8 (e >> 2) + 4 half + (e & 3).  W2p as ffn_planes."""
    W1 = np.asarray(W1, np.float32)
    w1q, w2p = ffn_planes(W1[:, acc_order_perm(W1.shape[1])], W2, scheme)
    return row_blocks(np.asarray(Wo, np.float32), scheme), w1q, w2p


def acc_order_perm(K: int) -> np.ndarray:
    """k order of a B operand that comes straight out of accumulator registers (csrc/ffn_fused.hip: the in-register LayerNorm): image
    column 16 ks + 8 h + e <- column 16 ks + 8 (e >> 2) + 4 h + (e & 3)."""
    k = np.arange(K)
    e, h, ks = k & 7, (k >> 3) & 1, k >> 4
    return 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3)


def outproj_q_planes(Wo: np.ndarray, Wq: np.ndarray, scheme=1):
    """Operand images of ctrlsim_outproj_ln_q (csrc/ffn_fused.hip, QP): -> (Wop, Wqp), eight W1-shaped blocks of 32 output columns each;
    Wqp in the k order of the LayerNorm output registers (acc_order_perm)."""
    Wq = np.asarray(Wq, np.float32)
    return row_blocks(np.asarray(Wo, np.float32), scheme), row_blocks(np.ascontiguousarray(Wq[:, acc_order_perm(Wq.shape[1])]), scheme)


def pack(dims: Dims, w: dict):
    """-> (flat float32 ndarray, names list, offsets int64 ndarray in floats)."""
    allw = dict(w)
    allw.update(fold(dims, w))
    # bf16x3 planes of every matrix that feeds the MFMA GEMM (2-D, K multiple of 16); stored as raw 16-bit words
    for k in list(allw.keys()):
        v = allw[k]
        if v.ndim == 2 and v.shape[1] % 16 == 0 and v.shape[1] >= 32 and (k.endswith("weight") or k.endswith(".w")) \
                and "embed_action" not in k and "embed_rtg_" not in k and "rtg_table" not in k and "embed_timestep" not in k and "embed_agent_id" not in k:
            for sch, suffix in PLANES_SUFFIX.items():      # both operand splits travel: the scheme is a run-time option
                try:
                    allw[k + suffix] = split3_planes(v, sch).reshape(-1).view(np.float32)
                except FloatingPointError:
                    if sch == 0:
                        raise                                  # fp16 planes out of range: that model runs with bf16 planes only
    # fused-FFN operand images of every transformer layer (post-LN block: linear1 -> ReLU -> linear2 -> +x -> LayerNorm)
    for k in list(w.keys()):
        if k.endswith(".linear1.weight"):
            pre = k[:-len(".linear1.weight")]
            for sch, suffix in PLANES_SUFFIX.items():
                try:
                    w1p, w2p = ffn_planes(np.asarray(w[k], np.float32), np.asarray(w[pre + ".linear2.weight"], np.float32), sch)
                except FloatingPointError:
                    if sch == 0:
                        raise
                    continue
                allw[pre + ".ffn#w1p" + suffix] = w1p.reshape(-1).view(np.float32)
                allw[pre + ".ffn#w2p" + suffix] = w2p.reshape(-1).view(np.float32)
    # the out-projection + LayerNorm in front of a feed-forward block as its leading product (two-fp16-plane scheme): decoder layers
    # multihead_attn.out_proj + norm2, encoder layers self_attn.out_proj + norm1
    for k in list(w.keys()):
        if k.endswith(".linear1.weight"):
            pre = k[:-len(".linear1.weight")]
            ok = pre + (".multihead_attn.out_proj.weight" if (pre + ".multihead_attn.out_proj.weight") in w else ".self_attn.out_proj.weight")
            if ok in w and np.asarray(w[ok]).shape == (256, 256) and np.asarray(w[k]).shape[1] == 256:
                try:
                    wop, w1q, _ = ffn_planes_pre(np.asarray(w[ok], np.float32), np.asarray(w[k], np.float32),
                                                 np.asarray(w[pre + ".linear2.weight"], np.float32), 1)
                except FloatingPointError:
                    continue
                allw[pre + ".ffn#wop" + PLANES_SUFFIX[1]] = wop.reshape(-1).view(np.float32)
                allw[pre + ".ffn#w1q" + PLANES_SUFFIX[1]] = w1q.reshape(-1).view(np.float32)
    # a decoder layer's self-attention out-projection + norm1 with the cross-attention query projection behind it (ctrlsim_outproj_ln_q)
    for k in list(w.keys()):
        if k.endswith(".multihead_attn.in_proj_weight"):
            pre = k[:-len(".multihead_attn.in_proj_weight")]
            ok = pre + ".self_attn.out_proj.weight"
            if ok in w and np.asarray(w[ok]).shape == (256, 256) and np.asarray(w[k]).shape == (768, 256):
                try:
                    wop, wqp = outproj_q_planes(np.asarray(w[ok], np.float32), np.asarray(w[k], np.float32)[:256], 1)
                except FloatingPointError:
                    continue
                allw[pre + ".selfq#wop" + PLANES_SUFFIX[1]] = wop.reshape(-1).view(np.float32)
                allw[pre + ".selfq#wqp" + PLANES_SUFFIX[1]] = wqp.reshape(-1).view(np.float32)
    # 32-column operand blocks of every attention in_proj (two-fp16-plane scheme: the row-stationary kernel)
    for k in list(w.keys()):
        if k.endswith("in_proj_weight") and np.asarray(w[k]).shape == (3 * 256, 256):
            try:
                allw[k + "#blk" + PLANES_SUFFIX[1]] = row_blocks(np.asarray(w[k], np.float32), 1).reshape(-1).view(np.float32)
            except FloatingPointError:
                pass                                               # out of the fp16 range: that model runs with bf16 planes
    names, offsets, chunks = [], [], []
    off = 0
    for k, v in allw.items():
        v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
        pad = (-off) % 64
        if pad:
            chunks.append(np.zeros(pad, np.float32))
            off += pad
        names.append(k)
        offsets.append(off)
        chunks.append(v)
        off += v.size
    return np.concatenate(chunks), names, np.asarray(offsets, np.int64)
