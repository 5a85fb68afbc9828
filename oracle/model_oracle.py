"""TEST INFRASTRUCTURE — CPU oracle of the CtRL-Sim model forward (not shipped, not a fallback).

A plain-PyTorch fp32 restatement of the reference's Encoder / MapEncoder / Decoder, written as
functional ops over an explicit weight dict (keys = the reference state_dict names, see
`ctrlsim_amd.weights.param_table`).  It follows, line by line in *behaviour*:

  modules/encoder.py:50-178      token embeddings, existence masking, (state, rtg, action) interleave,
                                 embed_ln, scene transformer-encoder over [polylines || initial states]
  modules/map_encoder.py:28-53   point MLP, single-seed 8-head attention pooling, norm1/map_feats/norm2,
                                 road-type MLP, fusion MLP, polyline validity
  modules/decoder.py:39-79       4 post-LN decoder layers (masked self-attn, cross-attn with
                                 memory_key_padding_mask, ReLU FFN), three MLP heads
  utils/train_utils.py:81-129    causal multi-agent mask, here as the closed form
                                 visible(i,j) <=> t_j<t_i or (t_j==t_i and ((a_j==a_i and k_j<=k_i) or k_j==0))
  torch.nn.TransformerEncoderLayer / TransformerDecoderLayer / MultiheadAttention (torch==2.2.0 pinned
  by the reference's environment.yml:28; un-vendored): post-LN, ReLU, eps=1e-5, additive float masks.

Pinned against the real reference modules (imported in the build container by
`oracle/gen_golden.py`) through the fixtures in tests/golden/model_*.npz.

It computes every head at every position with two dense passes' cost structure, i.e. the
reference's cost, which is why `bench.py` also uses it (via oracle/rollout_oracle.py) as
`cpu_baseline` kind "port".
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def as_torch_weights(w):
    return {k: (torch.from_numpy(v) if not torch.is_tensor(v) else v) for k, v in w.items()}


def causal_mask_closed_form(A: int, T: int, K: int = 3, state_index: int = 0, own_return: bool = False) -> torch.Tensor:
    """Boolean [L,L], True = visible; K = token types per agent-step: 3 CtRL-Sim, 2 IL, 1 Trajeglish (utils/train_utils.py:83-113
    with num_types = K); state_index 1 = Decision Transformer (token order rtg, state, action).  own_return =
    cfg.model.attend_own_return_action (utils/train_utils.py:114-129): of the EARLIER timesteps a token sees the state tokens and its
    own agent's tokens only — the other agents' past return / action tokens are hidden."""
    L = A * T * K
    i = torch.arange(L)
    t = i // (A * K)
    a = (i // K) % A
    k = i % K
    ti, tj = t[:, None], t[None, :]
    ai, aj = a[:, None], a[None, :]
    ki, kj = k[:, None], k[None, :]
    earlier = tj < ti
    if own_return:
        earlier = earlier & ((aj == ai) | (kj == state_index))
    return earlier | ((tj == ti) & (((aj == ai) & (kj <= ki)) | (kj == state_index)))


def _linear(x, w, name):
    return F.linear(x, w[name + ".weight"], w[name + ".bias"])


def _ln(x, w, name):
    return F.layer_norm(x, (x.shape[-1],), w[name + ".weight"], w[name + ".bias"], 1e-5)


def _mlp(x, w, name):  # utils/layers.py:6-19
    h = _linear(x, w, name + ".mlp.0")
    h = F.relu(_ln(h, w, name + ".mlp.1"))
    return _linear(h, w, name + ".mlp.3")


# CPU-baseline timing only (bench.py's cpu_baseline, tools/cpu_port_vs_reference.py): evaluate attention with torch's fused
# scaled_dot_product_attention — the op nn.MultiheadAttention itself calls in the reference (modules/decoder.py:16-20,
# encoder.py:42-46) — instead of the explicit softmax(Q K^T) V below.  Same function, other summation order (differences ~1e-6);
# the parity tests keep the explicit form they were pinned with.  Measured on the build container: 0.90 -> see
# profiles/r04_cpu_port_vs_reference.txt seconds per focal-group step against 0.53 for the unmodified reference.
FUSED_SDPA = False


def _mha(q_in, k_in, v_in, w, name, H, key_mask=None, attn_mask=None):
    """batch-first multi-head attention.  key_mask: bool [B,Lk] True = ignore.  attn_mask: bool [Lq,Lk]
    True = visible."""
    D = q_in.shape[-1]
    Wi, bi = w[name + ".in_proj_weight"], w[name + ".in_proj_bias"]
    q = F.linear(q_in, Wi[:D], bi[:D])
    k = F.linear(k_in, Wi[D:2 * D], bi[D:2 * D])
    v = F.linear(v_in, Wi[2 * D:], bi[2 * D:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    dh = D // H
    q = q.view(B, Lq, H, dh).transpose(1, 2)
    k = k.view(B, Lk, H, dh).transpose(1, 2)
    v = v.view(B, Lk, H, dh).transpose(1, 2)
    if FUSED_SDPA:
        bias = None
        if attn_mask is not None:
            bias = torch.zeros(Lq, Lk).masked_fill(~attn_mask, float("-inf"))[None, None]
        if key_mask is not None:
            kb = torch.zeros(B, 1, 1, Lk).masked_fill(key_mask[:, None, None, :], float("-inf"))
            bias = kb if bias is None else bias + kb
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias).transpose(1, 2).reshape(B, Lq, D)
        return _linear(o, w, name + ".out_proj")
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if attn_mask is not None:
        s = s.masked_fill(~attn_mask[None, None], float("-inf"))
    if key_mask is not None:
        s = s.masked_fill(key_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, D)
    return _linear(o, w, name + ".out_proj")


def map_encoder(w, road_points, road_types, H):
    """modules/map_encoder.py:34-53.  road_points [B,P,NP,3], road_types [B,P,8] -> [B,P,D], valid [B,P]."""
    pre = "encoder.map_encoder."
    rp = road_points.float()
    rt = road_types.float()
    B, P, NP, _ = rp.shape
    exist = rp[..., -1]
    seg_invalid = exist.sum(-1) == 0                         # [B,P]
    pts_mask = (1.0 - exist).bool().view(B * P, NP).clone()   # True = padded point
    pts_mask[:, 0][pts_mask.sum(-1) == NP] = False            # map_encoder.py:31
    feats = _mlp(rp, w, pre + "road_pts_encoder").view(B * P, NP, -1)
    type_feats = _mlp(rt, w, pre + "road_type_encoder").view(B * P, 1, -1)
    seeds = w[pre + "map_seeds"].view(1, 1, -1).expand(B * P, 1, -1)
    emb = _mha(seeds, feats, feats, w, pre + "road_pts_attn_layer", H, key_mask=pts_mask)
    emb = _ln(emb, w, pre + "norm1")
    emb2 = _ln(emb + _mlp(emb, w, pre + "map_feats"), w, pre + "norm2")
    emb2 = _mlp(torch.cat([emb2, type_feats], -1), w, pre + "road_road_type_encoder")
    return emb2.view(B, P, -1), ~seg_invalid


def embed_tokens(w, data, dims):
    """modules/encoder.py:50-153.  Returns stacked [B,L,D] (after embed_ln), initial-state embeddings
    [B,A,D] and initial existence [B,A] (bool)."""
    ag = data["agent_states"]
    B, A, T, _ = ag.shape
    exist = ag[..., -1:].transpose(1, 2)                              # [B,T,A,1]
    types = data["agent_types"].unsqueeze(2).expand(B, A, T, -1).transpose(1, 2)
    goals = data["goals"].unsqueeze(2).expand(B, A, T, -1).transpose(1, 2)[..., :dims.GOAL]
    states = torch.cat([ag[..., :-1].transpose(1, 2), types], -1).float()   # [B,T,A,12]
    actions = data["actions"].transpose(1, 2).long()                  # [B,T,A]
    rtgs = data["rtgs"].transpose(1, 2).long()                        # [B,T,A,3] (bins; continuous values for the DT variant)
    ts = data["timesteps"].transpose(1, 2).reshape(B, T, A).long()
    ids = torch.arange(A).view(1, 1, A).expand(B, T, A)

    ts_emb = F.embedding(ts, w["encoder.embed_timestep.weight"])
    id_emb = F.embedding(ids, w["encoder.embed_agent_id.weight"])
    s_emb = _mlp(states, w, "encoder.embed_state")
    g_emb = _mlp(goals.float(), w, "encoder.embed_goal")
    s_emb = _linear(torch.cat([s_emb, g_emb], -1), w, "encoder.embed_state_goal") + ts_emb + id_emb
    a_emb = F.embedding(actions, w["encoder.embed_action.weight"]) + ts_emb + id_emb
    if getattr(dims, "VARIANT", 0) == 3:                              # decision transformer: nn.Linear(1, D) on continuous RTGs
        rc = data["rtgs"].transpose(1, 2).float()                     # encoder.py:119-123
        r_emb = torch.cat([_linear(rc[..., c:c + 1], w, f"encoder.embed_rtg_{nm}")
                           for c, nm in enumerate(("goal", "veh", "road"))], -1)
    else:
        r_emb = torch.cat([
            F.embedding(rtgs[..., 0], w["encoder.embed_rtg_goal.weight"]),
            F.embedding(rtgs[..., 1], w["encoder.embed_rtg_veh.weight"]),
            F.embedding(rtgs[..., 2], w["encoder.embed_rtg_road.weight"])], -1)
    r_emb = _linear(r_emb, w, "encoder.embed_rtg") + ts_emb + id_emb
    ex = exist.float()
    # cfg.model.no_actions (encoder.py:129-130): the action embeddings are multiplied by zeros instead of the existence mask
    s_emb, a_emb, r_emb = s_emb * ex, a_emb * (torch.zeros_like(ex) if (getattr(dims, "FLAGS", 0) & 1) else ex), r_emb * ex
    init_emb = s_emb[:, 0]                                            # [B,A,D]
    init_exist = exist[:, 0, :, 0].bool()
    variant = getattr(dims, "VARIANT", 0)
    if variant == 1:                                                  # il: (state, action)           encoder.py:143-146
        stacked = torch.stack([s_emb, a_emb], dim=3).reshape(B, T * A * 2, -1)
    elif variant == 2:                                                # trajeglish: action tokens only  encoder.py:141-142
        stacked = a_emb.reshape(B, T * A, -1)
    elif variant == 3:                                                # decision transformer: (rtg, state, action)  encoder.py:137-140
        stacked = torch.stack([r_emb, s_emb, a_emb], dim=3).reshape(B, T * A * 3, -1)
    else:
        stacked = torch.stack([s_emb, r_emb, a_emb], dim=3).reshape(B, T * A * 3, -1)
    stacked = _ln(stacked, w, "encoder.embed_ln")
    return stacked, init_emb, init_exist


def _enc_layer(x, w, p, H, key_mask):
    x = _ln(x + _mha(x, x, x, w, p + ".self_attn", H, key_mask=key_mask), w, p + ".norm1")
    ff = _linear(F.relu(_linear(x, w, p + ".linear1")), w, p + ".linear2")
    return _ln(x + ff, w, p + ".norm2")


def _dec_layer(x, mem, w, p, H, tgt_mask, mem_key_mask):
    x = _ln(x + _mha(x, x, x, w, p + ".self_attn", H, attn_mask=tgt_mask), w, p + ".norm1")
    x = _ln(x + _mha(x, mem, mem, w, p + ".multihead_attn", H, key_mask=mem_key_mask), w, p + ".norm2")
    ff = _linear(F.relu(_linear(x, w, p + ".linear1")), w, p + ".linear2")
    return _ln(x + ff, w, p + ".norm3")


_MASK_CACHE = {}


def forward(w, data, dims, return_hidden=False):
    """CtRLSim.forward (models/ctrl_sim.py:41-45) on a dict of batched tensors.
    data: agent_states [B,A,T,8], agent_types [B,A,5], goals [B,A,5], actions [B,A,T], rtgs [B,A,T,3],
          timesteps [B,A,T,1], road_points [B,P,NP,3], road_types [B,P,8]."""
    H = dims.H
    stacked, init_emb, init_exist = embed_tokens(w, data, dims)
    flags = getattr(dims, "FLAGS", 0)
    seg = None
    if flags & 2:                                                     # use_map False (encoder.py:168-170): the initial states alone
        src, pad = init_emb, ~init_exist
    else:
        seg, valid = map_encoder(w, data["road_points"], data["road_types"], H)
        if flags & 4:                                                 # encode_initial_state False (encoder.py:163-166): the polylines alone
            src, pad = seg, ~valid
        else:
            src = torch.cat([seg, init_emb], 1)
            pad = ~torch.cat([valid, init_exist], 1)
    mem = src
    for i in range(dims.NE):
        mem = _enc_layer(mem, w, f"encoder.transformer_encoder.layers.{i}", H, pad)
    B, A, T = data["agent_states"].shape[:3]
    variant = getattr(dims, "VARIANT", 0)
    K = {0: 3, 1: 2, 2: 1, 3: 3}[variant]                             # decoder.py:29-35
    own = bool(getattr(dims, "MASK_OWN", False))
    key = (A, T, K, variant == 3, own)
    if key not in _MASK_CACHE:
        _MASK_CACHE[key] = causal_mask_closed_form(A, T, K, 1 if variant == 3 else 0, own)
    tgt_mask = _MASK_CACHE[key]
    x = stacked
    for i in range(dims.ND):
        x = _dec_layer(x, mem, w, f"decoder.transformer_decoder.layers.{i}", H, tgt_mask, pad)
    if variant:                                                       # decoder.py:55-64: one head; DT reads token 1 (its state
        out = x.view(B, T * A, K, -1)                                 # token), IL / Trajeglish token 0
        act = _mlp(out[:, :, 1 if variant == 3 else 0], w, "decoder.predict_action").view(B, T, A, -1).permute(0, 2, 1, 3)
        return {"action_preds": act}
    out = x.view(B, T * A, 3, -1)
    act = _mlp(out[:, :, 1], w, "decoder.predict_action").view(B, T, A, -1).permute(0, 2, 1, 3)
    rtg = _mlp(out[:, :, 0], w, "decoder.predict_rtg").view(B, T, A, -1).permute(0, 2, 1, 3)
    fut = _mlp(out[:, :, 2], w, "decoder.predict_future_states").view(B, T, A, -1).permute(0, 2, 1, 3)
    preds = {"action_preds": act, "rtg_preds": rtg, "state_preds": fut}
    if return_hidden:
        preds["stacked_embeddings"] = stacked
        preds["road_seg_emb"] = seg
        preds["encoder_embeddings"] = mem
        preds["decoder_out"] = x
    return preds
