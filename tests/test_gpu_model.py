"""GPU parity of the HIP forward (pass 1 / pass 2) against the CPU oracle and the reference's golden logits."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import golden, cfg_of  # noqa: E402
from ctrlsim_amd import spec, weights, _lib  # noqa: E402
from ctrlsim_amd.engine import HipModel, ctx_from_reference_layout  # noqa: E402
import model_oracle as mo  # noqa: E402
import synth_inputs  # noqa: E402
from gpu_utils import DEV  # noqa: E402

TOL = 1e-4   # north-star tolerance on fp32 logits; observed error is ~1e-5 (fp32 reassociation of folded weights)


def _run_both_passes(model, d, inp, Tq, new_bins):
    """-> rtg logits [B,A,R*C] (pass 1), action logits [B,A,V] (pass 2 after writing new_bins [B,A,3] at row ti)."""
    B = inp["agent_states"].shape[0]
    cb = ctx_from_reference_layout(d, inp, Tq, DEV)
    cb.slot_gid.copy_(torch.arange(d.A, dtype=torch.int32, device=DEV).expand(B, d.A))
    ws = torch.empty(model.workspace_bytes(B, Tq), dtype=torch.uint8, device=DEV)
    rtg = torch.empty(B, d.A, d.R * d.C, device=DEV)
    act = torch.empty(B, d.A, d.V, device=DEV)
    seg = torch.empty(B, d.P, d.D, device=DEV)
    lib, st = _lib.lib(), _lib.stream_ptr()
    _lib.check(lib.ctrlsim_dt_forward_pass1(model.handle, B, Tq, C.byref(cb.struct), ws.data_ptr(), rtg.data_ptr(),
                                            seg.data_ptr(), st), "pass1")
    Tmax, t = 90, 40
    hist_rtg = torch.zeros(B, d.A, Tmax, 3, dtype=torch.int32, device=DEV)
    hist_rtg[:, :, t] = torch.from_numpy(new_bins.astype(np.int32)).to(DEV)
    ctx_scn = torch.arange(B, dtype=torch.int32, device=DEV)
    _lib.check(lib.ctrlsim_dt_forward_pass2(model.handle, B, Tq, t, d.A, Tmax, C.byref(cb.struct), ctx_scn.data_ptr(),
                                            hist_rtg.data_ptr(), ws.data_ptr(), act.data_ptr(), 0, st), "pass2")
    torch.cuda.synchronize()
    return rtg.cpu().numpy(), act.cpu().numpy(), seg.cpu().numpy()


@pytest.mark.parametrize("kind,B", [("tiny", 3), ("loop", 2), ("full", 2), ("wide", 1)])
def test_forward_matches_oracle(kind, B):
    """(wide: A = 64 slots, P = 512 polylines in ONE context, L = 6144 token rows — the non-reference "wide context" of SURVEY.md section 8(d))"""
    cfg = cfg_of(kind)
    d = spec.Dims(cfg)
    w = weights.generate(d, 0)
    model = HipModel(cfg, w, DEV)
    tw = mo.as_torch_weights(w)
    for seed, Tq in ((11, d.T), (12, max(1, d.T // 3)), (13, 1))[:1 if kind == "wide" else 3]:     # (wide: the full window only — the oracle's two
        inp = synth_inputs.random_context(d, seed, B=B, t_fill=Tq, n_agents=d.A - 1, n_polys=d.P - 1)  #  6144-row forwards take ~10 s of CPU each)
        ti = Tq - 1
        rs = np.random.RandomState(seed)
        new_bins = rs.randint(0, d.R, (B, d.A, 3))
        rtg, act, seg = _run_both_passes(model, d, inp, Tq, new_bins)
        with torch.no_grad():
            o1 = mo.forward(tw, synth_inputs.to_torch(inp), d, return_hidden=True)
            inp2 = {k: v.copy() for k, v in inp.items()}
            inp2["rtgs"][:, :, ti] = new_bins
            o2 = mo.forward(tw, synth_inputs.to_torch(inp2), d)
        np.testing.assert_allclose(seg, o1["road_seg_emb"].numpy(), atol=TOL, rtol=0)
        np.testing.assert_allclose(rtg, o1["rtg_preds"][:, :, ti].numpy(), atol=TOL, rtol=0)
        np.testing.assert_allclose(act, o2["action_preds"][:, :, ti].numpy(), atol=TOL, rtol=0)
        # pass 1 must not depend on the placeholder, pass 2 must (sanity of the reuse argument)
        assert np.abs(o2["rtg_preds"][:, :, ti].numpy() - o1["rtg_preds"][:, :, ti].numpy()).max() < 1e-6
        assert np.abs(o2["action_preds"][:, :, ti].numpy() - o1["action_preds"][:, :, ti].numpy()).max() > 1e-4


def test_forward_matches_reference_golden_logits():
    """tests/golden/model_full.npz holds logits of the UNMODIFIED reference Encoder/Decoder (oracle/gen_golden.py)."""
    cfg = cfg_of("full")
    d = spec.Dims(cfg)
    model = HipModel(cfg, weights.generate(d, 0), DEV)
    g = golden("model_full")
    for seed in (1, 2):
        _, t_fill, n_ag, n_pl = [int(v) for v in g[f"s{seed}_recipe"]]
        inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
        ti = t_fill - 1
        bins = inp["rtgs"][:, :, ti].astype(np.int64)          # pass 2 with the bins already in the input
        rtg, act, _ = _run_both_passes(model, d, inp, t_fill, bins)
        np.testing.assert_allclose(rtg[0], g[f"s{seed}_rtg_logits"], atol=TOL, rtol=0)
        np.testing.assert_allclose(act[0], g[f"s{seed}_action_logits"], atol=TOL, rtol=0)


@pytest.mark.parametrize("split", ["f16x3", "bf16x6"])
def test_forward_matches_reference_logits_at_trained_like_weights(split):
    """Round 5 (round-4 review, missing #3 / weak #1): parity beyond the random-init regime.  tests/golden/model_trained.npz holds the
    logits of the UNMODIFIED reference Encoder / Decoder at full dims with trained-like weights (weights.generate_trained_like: LayerNorm
    gains in [0.5, 2], matrix rows / columns rescaled, embedding rows over three decades, peaked attention, head gain 15 -> |logit| ~ 30)
    and the float64 evaluation of the same network.  Both operand splits must stay within 1e-5 of max |logit| of the reference's float32
    logits (relative bound: the error of a logit scales with the head gain); the fixture's own float32-vs-float64 distance is printed
    next to ours — the reference itself is 2-3e-6 of max |logit| away from exact arithmetic here."""
    from ctrlsim_amd.engine import RolloutEngine  # noqa: F401  (bind helper below)
    cfg = cfg_of("full")
    d = spec.Dims(cfg)
    g = golden("model_trained")
    lib = _lib.lib()
    for seed in (1, 2, 3):
        _, t_fill, n_ag, n_pl, wseed = [int(v) for v in g[f"s{seed}_recipe"]]
        model = HipModel(cfg, weights.generate_trained_like(d, wseed), DEV)
        inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
        ti = t_fill - 1
        bins = inp["rtgs"][:, :, ti].astype(np.int64)
        guard = torch.zeros(2, dtype=torch.int32, device=DEV)
        _lib.check(lib.ctrlsim_bind(1 if split == "f16x3" else 0, guard.data_ptr()))
        try:
            rtg, act, _ = _run_both_passes(model, d, inp, t_fill, bins)
        finally:
            lib.ctrlsim_bind(1, None)
        assert guard.tolist() == [0, 0], "non-finite rows at trained-like weights: the fp16 planes overflowed"
        for ours, nm in ((rtg, "rtg_logits"), (act, "action_logits")):
            ref, f64 = g[f"s{seed}_{nm}"][:n_ag], g[f"s{seed}_{nm}_f64"][:n_ag]
            scale = np.abs(ref).max()
            err, ref_err = np.abs(ours[0, :n_ag] - ref).max(), np.abs(ref - f64).max()
            print(f"{split} s{seed} {nm}: max|logit| {scale:.1f}, ours vs reference {err / scale:.2e}, ours vs float64 "
                  f"{np.abs(ours[0, :n_ag] - f64).max() / scale:.2e}, reference vs float64 {ref_err / scale:.2e} (relative)")
            assert scale > 15.0
            assert err <= 1e-5 * scale, (split, seed, nm, err, scale)


@pytest.mark.parametrize("fused", [0, 1, 2, 3])
def test_forward_golden_logits_at_every_setting_of_the_out_projection_fusion(fused):
    """Option 3 (csrc/common.h: OPT_FFN_FUSED): 0 = separate Linear kernels, 1 = the feed-forward block as one kernel, 2 (default) = with the
    attention out-projection + residual + LayerNorm in front of it as its leading product (ctrlsim_ffn_fused_pre), 3 = and the self-attention
    out-projection + norm1 + cross-attention query projection as one kernel (ctrlsim_outproj_ln_q).  Every setting against the UNMODIFIED
    reference's logits (tests/golden/model_full.npz, model_trained.npz), bound per call so that the process default stays untouched."""
    cfg = cfg_of("full")
    d = spec.Dims(cfg)
    lib = _lib.lib()
    n_opt = lib.ctrlsim_option_count()
    vals = (C.c_int * n_opt)(*([-1] * n_opt))
    vals[3] = fused
    guard = torch.zeros(2, dtype=torch.int32, device=DEV)
    for fixture, gen, seeds in (("model_full", lambda ws: weights.generate(d, 0), (1,)),
                                ("model_trained", lambda ws: weights.generate_trained_like(d, ws), (1, 2))):
        g = golden(fixture)
        for seed in seeds:
            rec = [int(v) for v in g[f"s{seed}_recipe"]]
            t_fill, n_ag, n_pl = rec[1], rec[2], rec[3]
            model = HipModel(cfg, gen(rec[4] if len(rec) > 4 else 0), DEV)
            inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
            bins = inp["rtgs"][:, :, t_fill - 1].astype(np.int64)
            _lib.check(lib.ctrlsim_bind(1, guard.data_ptr()))
            _lib.check(lib.ctrlsim_bind_options(vals))
            try:
                assert lib.ctrlsim_get_option(3) == fused
                rtg, act, _ = _run_both_passes(model, d, inp, t_fill, bins)
            finally:
                lib.ctrlsim_bind_options(None)
                lib.ctrlsim_bind(1, None)
            assert guard.tolist() == [0, 0]
            for ours, nm in ((rtg, "rtg_logits"), (act, "action_logits")):
                ref = g[f"s{seed}_{nm}"][:n_ag]
                scale = max(np.abs(ref).max(), 10.0)
                assert np.abs(ours[0, :n_ag] - ref).max() <= 1e-5 * scale, (fused, fixture, seed, nm)


def test_forward_f32_mfma_kernels_selectable():
    """ctrlsim_set_option(0/1, 0) routes every Linear / attention through the f32-input MFMA kernels (separate LayerNorm
    kernel); the logits must agree with the default split-bf16 path to well inside the tolerance."""
    cfg = cfg_of("loop")
    d = spec.Dims(cfg)
    model = HipModel(cfg, weights.generate(d, 0), DEV)
    inp = synth_inputs.random_context(d, 5, B=2, t_fill=d.T, n_agents=d.A - 1, n_polys=d.P - 1)
    bins = np.random.RandomState(5).randint(0, d.R, (2, d.A, 3))
    lib = _lib.lib()
    a = _run_both_passes(model, d, inp, d.T, bins)
    try:
        lib.ctrlsim_set_option(0, 0); lib.ctrlsim_set_option(1, 0)
        b = _run_both_passes(model, d, inp, d.T, bins)
    finally:
        lib.ctrlsim_set_option(0, 1); lib.ctrlsim_set_option(1, 1)
    for x, y in zip(a, b):
        np.testing.assert_allclose(x, y, atol=TOL, rtol=0)


def _run_compact(model, d, inp, Tq, new_bins, Actx):
    """The same two passes on COMPACT contexts of Actx slots: the leading Actx slots of the reference layout, the last of them
    a padded slot that stands for all d.A - (Actx - 1) padded ones.  -> logits of the Actx - 1 regular slots."""
    B = inp["agent_states"].shape[0]
    sub = {k: (v[:, :Actx] if k in ("agent_states", "agent_types", "goals", "actions", "rtgs", "timesteps", "moving_agent_mask")
               else v) for k, v in inp.items()}
    cb = ctx_from_reference_layout(d, sub, Tq, DEV)
    Ar = Actx - 1 if Actx < d.A else d.A
    gid = torch.arange(Actx, dtype=torch.int32, device=DEV).expand(B, Actx).contiguous()
    cb.slot_gid.view(-1)[:B * Actx] = gid.reshape(-1)
    ws = torch.empty(model.workspace_bytes(B, Tq, Actx), dtype=torch.uint8, device=DEV)
    rtg = torch.empty(B, Ar, d.R * d.C, device=DEV)
    act = torch.empty(B, Ar, d.V, device=DEV)
    lib, st = _lib.lib(), _lib.stream_ptr()
    _lib.check(lib.ctrlsim_dt_forward_pass1_a(model.handle, B, Tq, Actx, C.byref(cb.struct), ws.data_ptr(), rtg.data_ptr(),
                                              None, st), "pass1_a")
    Tmax, t = 90, 40
    hist_rtg = torch.zeros(B, d.A, Tmax, 3, dtype=torch.int32, device=DEV)
    hist_rtg[:, :, t] = torch.from_numpy(new_bins.astype(np.int32)).to(DEV)
    ctx_scn = torch.arange(B, dtype=torch.int32, device=DEV)
    _lib.check(lib.ctrlsim_dt_forward_pass2_a(model.handle, B, Tq, Actx, t, d.A, Tmax, C.byref(cb.struct), ctx_scn.data_ptr(),
                                              hist_rtg.data_ptr(), ws.data_ptr(), act.data_ptr(), 0, st), "pass2_a")
    torch.cuda.synchronize()
    return rtg.cpu().numpy(), act.cpu().numpy()


@pytest.mark.parametrize("kind,n_agents,Actx,Tqs", [("loop", 3, 4, (8, 3, 1)), ("loop", 2, 4, (8,)), ("full", 7, 8, (32, 5)),
                                                     ("full", 10, 12, (32,)), ("full", 3, 4, (32, 1)), ("full", 19, 20, (32,))])
def test_compact_contexts_equal_plain_contexts(kind, n_agents, Actx, Tqs):
    """Padded agent slots are evaluated once, as one representative slot whose keys carry the multiplicity of the padded slots
    it stands for (forward.hip: Shape; attention_bf16x6.hip).  Logits of the live slots must equal the plain 24-slot (6-slot)
    evaluation of the same context — which the other tests pin to the reference — to fp32 round-off."""
    cfg = cfg_of(kind)
    d = spec.Dims(cfg)
    w = weights.generate(d, 0)
    model = HipModel(cfg, w, DEV)
    for seed, Tq in zip((21, 22, 23), Tqs):
        inp = synth_inputs.random_context(d, seed, B=2, t_fill=Tq, n_agents=n_agents, n_polys=d.P - 1)
        new_bins = np.random.RandomState(seed).randint(0, d.R, (2, d.A, 3))
        rtg, act, _ = _run_both_passes(model, d, inp, Tq, new_bins)
        rtg_c, act_c = _run_compact(model, d, inp, Tq, new_bins, Actx)
        Ar = Actx - 1
        np.testing.assert_allclose(rtg_c[:, :n_agents], rtg[:, :n_agents], atol=2e-5, rtol=0)
        np.testing.assert_allclose(act_c[:, :n_agents], act[:, :n_agents], atol=2e-5, rtol=0)
        assert np.isfinite(rtg_c).all() and rtg_c.shape[1] == Ar


def _motion_data(inp):
    return {"agent": {k: inp[k] for k in ("agent_states", "agent_types", "goals", "actions", "rtgs", "timesteps")},
            "map": {k: inp[k] for k in ("road_points", "road_types")}}


def test_forward_full_return_contract_matches_reference_and_oracle():
    """`CtRLSim.forward(data)` is the reference's contract (models/ctrl_sim.py:41-45, decoder.py:52-77): all three heads on every
    token, [B,A,T,.].  tests/golden/model_tiny.npz holds those tensors from the UNMODIFIED reference (including state_preds of
    the predict_future_states head); at full dims the oracle (pinned to the reference on the queried slices) is the checker."""
    from ctrlsim_amd.models import CtRLSim
    for kind in ("tiny", "full"):
        cfg = cfg_of(kind)
        d = spec.Dims(cfg)
        w = weights.generate(d, 0)
        model = CtRLSim(cfg, w, device=DEV)
        g = golden(f"model_{kind}")
        for seed in (1, 2):
            _, t_fill, n_ag, n_pl = [int(v) for v in g[f"s{seed}_recipe"]]
            inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
            out = model(_motion_data(inp))
            assert set(out) == {"action_preds", "rtg_preds", "state_preds"}
            assert out["state_preds"].shape == (1, d.A, d.T, 2 * d.T)
            if kind == "tiny":
                for k in out:
                    np.testing.assert_allclose(out[k].cpu().numpy(), g[f"s{seed}_{k}"], atol=TOL, rtol=0, err_msg=k)
            else:
                ti = t_fill - 1
                np.testing.assert_allclose(out["action_preds"][0, :, ti].cpu().numpy(), g[f"s{seed}_action_logits"], atol=TOL, rtol=0)
                np.testing.assert_allclose(out["rtg_preds"][0, :, ti].cpu().numpy(), g[f"s{seed}_rtg_logits"], atol=TOL, rtol=0)
                with torch.no_grad():
                    ref = mo.forward(mo.as_torch_weights(w), synth_inputs.to_torch(inp), d)
                for k in out:
                    np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), atol=TOL, rtol=0, err_msg=k)
            # the sliced two-pass call agrees with the full contract at the queried step
            ti = t_fill - 1
            sl = model(_motion_data(inp), token_index=ti)
            np.testing.assert_allclose(sl["rtg_preds"].cpu().numpy(), out["rtg_preds"][:, :, ti].cpu().numpy(), atol=TOL, rtol=0)
            np.testing.assert_allclose(sl["action_preds"].cpu().numpy(), out["action_preds"][:, :, ti].cpu().numpy(), atol=TOL, rtol=0)
