"""CPU tests: the oracle (oracle/*.py, oracle/sim_oracle.c) is pinned against golden vectors that were
produced by running the reference itself (oracle/gen_golden.py) — SURVEY.md §8c G1-G10."""
import numpy as np
import pytest
import torch

from helpers import golden, cfg_of
from ctrlsim_amd import spec, weights, scenarios
import features_oracle as fo
import model_oracle as mo
import synth_inputs
import sim_libs
import rollout_oracle


@pytest.fixture(scope="module", autouse=True)
def _build():
    sim_libs.build_oracle()


def test_weight_generator_matches_reference_parameter_count():
    d = spec.Dims(cfg_of("full"))
    assert weights.num_params(d) == 8285762            # SURVEY.md §8a M7 (probe of the reference modules)
    w = weights.generate(d, 0)
    w2 = weights.generate(d, 0)
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    assert abs(float(w["encoder.embed_action.weight"].std()) - 0.02) < 1e-3


def test_mask_closed_form_fraction():
    g = golden("model_full")
    cm = mo.causal_mask_closed_form(24, 32, 3)
    assert abs(float(cm.float().mean()) - float(g["mask_visible_fraction"])) < 1e-12   # 49.5 %


@pytest.mark.parametrize("kind", ["tiny", "full"])
def test_model_oracle_matches_reference(kind):
    cfg = cfg_of(kind)
    d = spec.Dims(cfg)
    tw = mo.as_torch_weights(weights.generate(d, 0))
    g = golden(f"model_{kind}")
    for seed in (1, 2):
        _, t_fill, n_ag, n_pl = [int(v) for v in g[f"s{seed}_recipe"]]
        inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
        with torch.no_grad():
            out = mo.forward(tw, synth_inputs.to_torch(inp), d)
        if kind == "tiny":
            for k in ("action_preds", "rtg_preds", "state_preds"):
                np.testing.assert_allclose(out[k].numpy(), g[f"s{seed}_{k}"], atol=2e-5, rtol=0)
        else:
            ti = t_fill - 1
            np.testing.assert_allclose(out["action_preds"][0, :, ti].numpy(), g[f"s{seed}_action_logits"], atol=2e-5, rtol=0)
            np.testing.assert_allclose(out["rtg_preds"][0, :, ti].numpy(), g[f"s{seed}_rtg_logits"], atol=2e-5, rtol=0)


def test_model_oracle_matches_reference_at_trained_like_weights():
    """Round 5: the oracle at trained-like weights (weights.generate_trained_like) against the UNMODIFIED reference modules' logits
    (tests/golden/model_trained.npz).  Two float32 evaluations of one network differ by their rounding noise, which here is larger than
    at random init (|logit| ~ 30, peaked attention): the bound is relative to max |logit| and the fixture's own float32-vs-float64
    distance is checked to be of the same order — the regime is well conditioned, the oracle is not just "close by luck"."""
    cfg = cfg_of("full")
    d = spec.Dims(cfg)
    g = golden("model_trained")
    for seed in (1, 3):
        _, t_fill, n_ag, n_pl, wseed = [int(v) for v in g[f"s{seed}_recipe"]]
        tw = mo.as_torch_weights(weights.generate_trained_like(d, wseed))
        inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
        with torch.no_grad():
            out = mo.forward(tw, synth_inputs.to_torch(inp), d)
        ti = t_fill - 1
        for k, nm in (("action_preds", "action_logits"), ("rtg_preds", "rtg_logits")):
            ref, f64 = g[f"s{seed}_{nm}"][:n_ag], g[f"s{seed}_{nm}_f64"][:n_ag]
            scale = np.abs(ref).max()
            assert scale > 15.0                                                   # sharp logits
            assert np.abs(ref - f64).max() < 1e-5 * scale                         # the reference's float32 is near exact arithmetic here
            assert np.abs(out[k][0, :n_ag, ti].numpy() - ref).max() < 1e-5 * scale
        ra = g[f"s{seed}_action_logits"][:n_ag]
        assert (ra - ra.mean(0, keepdims=True)).std() > 0.3                       # and the logits DO depend on the vehicle / scene


@pytest.mark.parametrize("tag,kind,n_ag,n_pl,extent", [("small", "loop", 10, 20, 45.0), ("full", "full", 30, 260, 70.0),
                                                       ("wide", "full", 64, 512, 70.0)])
def test_feature_oracle_matches_reference(tag, kind, n_ag, n_pl, extent):
    cfg = cfg_of(kind)
    d = spec.Dims(cfg)
    w = cfg.dataset.waymo
    g = golden("features")
    scn = scenarios.make_scenario(11, 0, n_agents=n_ag, n_polylines=n_pl, n_points=d.NP, extent=extent)
    b = synth_inputs.synth_policy_buffers(scn, cfg, seed=5)
    buf = fo.PolicyBuffers(scn.N, cfg.nocturne.steps)
    for k in ("states", "types", "actions", "rtgs", "goals", "timesteps"):
        getattr(buf, k)[:] = b[k]
    n_checked = 0
    for t in (0, 3, d.T + 5):
        groups, dead = fo.build_contexts(buf, w, t, list(scn.eval_order), scn.road_points.astype(np.float64), scn.road_types)
        assert [gr["focal"] for gr in groups] == list(g[f"{tag}_t{t}_focals"])
        for gi, gr in enumerate(groups):
            pre = f"{tag}_t{t}_g{gi}_"
            assert gr["ids"] == list(g[pre + "ids"])
            assert gr["members"] == list(g[pre + "members"])
            if pre + "agent_states" in g:
                dt = gr["data"]
                # bit-exact: same float64 op sequence as the reference's NumPy path
                assert np.array_equal(dt["agent_states"][0], g[pre + "agent_states"])
                assert np.array_equal(dt["goals"][0], g[pre + "goals"])
                assert np.array_equal(dt["actions"][0], g[pre + "actions"])
                assert np.array_equal(dt["rtgs"][0], g[pre + "rtgs"])
                assert np.array_equal(dt["agent_types"][0], g[pre + "types"])
                assert np.array_equal(dt["timesteps"][0], g[pre + "timesteps"])
                assert np.array_equal(dt["road_types"][0], g[pre + "road_types"])
                rp = dt["road_points"][0]
                assert np.array_equal(rp if tag == "small" else rp[:, ::25], g[pre + "road_points"])
                np.testing.assert_allclose(rp.sum(axis=(1, 2)), g[pre + "road_points_sum"], rtol=1e-12)
                n_checked += 1
    assert n_checked >= 3


def test_discretisation_round_trips_and_placeholders():
    w = cfg_of("full").dataset.waymo
    tok = np.arange(1000)
    assert np.array_equal(fo.discretize_actions(fo.undiscretize_actions(tok, w), w), tok)
    bins = np.stack([np.arange(350)] * 3, -1)
    assert np.array_equal(fo.discretize_rtgs(fo.normalize_rtgs(fo.undiscretize_rtgs(bins, w), w), w), bins)
    assert fo.discretize_actions(np.zeros((1, 2)), w)[0] == spec.ZERO_ACTION_TOKEN        # half-to-even
    assert tuple(fo.discretize_rtgs(fo.normalize_rtgs(np.zeros((1, 3)), w), w)[0]) == spec.ZERO_RTG_BINS


def test_sampling_oracle_matches_reference():
    cfg = cfg_of("full")
    d = spec.Dims(cfg)
    w = cfg.dataset.waymo
    g = golden("sampling")
    assert tuple(g["multinomial_agreement"]) == (32, 32)
    n = g["rtg_logits"].shape[0]
    for ti, tl in enumerate(g["tilts"]):
        tilt = fo.tilt_logits(*tl, w)
        for i in range(n):
            noise = lambda head, m, i=i: weights.exp_noise(9, 0, 0, i, head, m)
            bins = rollout_oracle.sample_rtg(torch.from_numpy(g["rtg_logits"][i]), tilt, d.R, d.C, noise)
            assert bins == list(g[f"rtg_bins_tilt{ti}"][i])
    for tag, temp, nuc in (("t1", 1.0, False), ("t15", 1.5, False), ("nuc", 1.0, True), ("nuc_t07", 0.7, True)):
        for i in range(n):
            noise = lambda head, m, i=i: weights.exp_noise(9, 0, 0, i, head, m)
            tok = rollout_oracle.sample_action(torch.from_numpy(g["act_logits"][i]), temp, nuc, 0.8, noise)
            assert tok == int(g[f"act_tok_{tag}"][i])


def _run_scripted(sim_cls, g):
    sim = sim_cls(g["L"], g["W"], g["x"], g["y"], g["h"], g["v"], g["segs"])
    steps, n = g["acts"].shape[:2]
    traj = np.zeros((steps + 1, n, 6), np.float32); cv = np.zeros((steps + 1, n), np.uint8); ce = cv.copy()
    traj[0], cv[0], ce[0] = sim.state()
    for t in range(steps):
        for i in range(n):
            sim.set_action(i, g["acts"][t, i, 0], g["acts"][t, i, 1])
        sim.step(0.1)
        traj[t + 1], cv[t + 1], ce[t + 1] = sim.state()
    sim.close()
    return traj, cv, ce


def test_sim_oracle_bit_exact_vs_reference_physics_fixture():
    g = golden("physics")
    traj, cv, ce = _run_scripted(sim_libs.OracleSim, g)
    assert np.array_equal(traj, g["traj"])                      # float32 bit-exact (contact-free)
    assert np.array_equal(cv, g["coll_veh"]) and np.array_equal(ce, g["coll_edge"])
    assert g["coll_edge"].sum() > 0
    # brake-to-zero car is at rest, clamped car is at 50 m/s
    assert g["traj"][-1, 0, 3] == 0.0 and abs(g["traj"][-1, 1, 3] - 50.0) < 1e-4


@pytest.mark.skipif(not sim_libs.RefSim.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_sim_oracle_bit_exact_vs_live_reference_physics():
    g = golden("physics")
    a = _run_scripted(sim_libs.RefSim, g)
    b = _run_scripted(sim_libs.OracleSim, g)
    assert np.array_equal(a[0], g["traj"])
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def _contact_case(g, k):
    return {key: g[f"c{k}_{key}"] for key in ("L", "W", "x", "y", "h", "v", "acts", "segs")}


def _run_scripted_body(sim_cls, sc):
    sim = sim_cls(sc["L"], sc["W"], sc["x"], sc["y"], sc["h"], sc["v"], sc["segs"])
    steps, n = sc["acts"].shape[:2]
    traj = np.zeros((steps + 1, n, 6), np.float32); cv = np.zeros((steps + 1, n), np.uint8)
    body = np.zeros((steps + 1, n, 6), np.float32)
    traj[0], cv[0], _ = sim.state(); body[0] = sim.body()
    for t in range(steps):
        for i in range(n):
            sim.set_action(i, sc["acts"][t, i, 0], sc["acts"][t, i, 1])
        sim.step(0.1)
        traj[t + 1], cv[t + 1], _ = sim.state(); body[t + 1] = sim.body()
    sim.close()
    return traj, cv, body


def test_sim_oracle_contacts_bit_exact_vs_reference_fixture():
    """tests/golden/contacts.npz: vehicles running into each other through the REAL FreeCar + Box2D (manifolds, warm-started
    sequential impulses, block solver, position correction).  The C restatement must reproduce positions, headings,
    speeds AND Box2D body velocities bit for bit (float32), through and after the collisions."""
    g = golden("contacts")
    for k in range(int(g["n_cases"])):
        traj, cv, body = _run_scripted_body(sim_libs.OracleSim, _contact_case(g, k))
        assert g[f"c{k}_coll_veh"].sum() > 0
        assert np.array_equal(cv, g[f"c{k}_coll_veh"]), k
        assert np.array_equal(traj.view(np.int32), g[f"c{k}_traj"].view(np.int32)), k
        assert np.array_equal(body.view(np.int32), g[f"c{k}_body"].view(np.int32)), k


@pytest.mark.skipif(not sim_libs.RefSim.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_sim_oracle_contacts_bit_exact_vs_live_reference():
    """Fresh random encounters against the real Box2D, live: two-car collisions (alone or several per world), pile-ups of
    8-16 cars and lots of 32-48 cars that overlap from the start — all bit-exact (the order in which contacts of one step
    are created comes from the restated b2DynamicTree)."""
    import gen_golden
    n_coll = 0
    same = lambda x, y: np.array_equal(x.view(x.dtype if x.dtype == np.uint8 else np.int32),
                                       y.view(y.dtype if y.dtype == np.uint8 else np.int32))
    for kind in ("headon", "tbone", "pairs"):
        for seed in range(20, 32):
            sc = gen_golden.contact_scene(kind, seed)
            a = _run_scripted_body(sim_libs.RefSim, sc)
            b = _run_scripted_body(sim_libs.OracleSim, sc)
            n_coll += int(a[1].sum())
            assert all(same(x, y) for x, y in zip(a, b)), (kind, seed)
    assert n_coll > 100
    for kind, seeds in (("crowd", range(20, 32)), ("dense", range(20, 26))):
        for seed in seeds:
            sc = gen_golden.contact_scene(kind, seed)
            a = _run_scripted_body(sim_libs.RefSim, sc)
            b = _run_scripted_body(sim_libs.OracleSim, sc)
            assert a[1].sum() > 20
            assert all(same(x, y) for x, y in zip(a, b)), (kind, seed)


def test_collision_oracle_matches_reference_geometry():
    g = golden("collision")
    geo = sim_libs.oracle_geo()
    for i in range(len(g["segs"])):
        a = np.ascontiguousarray(g["boxes"][i, 0]); b = np.ascontiguousarray(g["boxes"][i, 1])
        assert geo.orageo_poly_poly(a, 4, b, 4) == g["poly_poly"][i]
        assert geo.orageo_poly_seg(a, 4, np.ascontiguousarray(g["segs"][i])) == g["poly_seg"][i]


def test_collision_oracle_reference_kats():
    """Known-answer cases held by the reference's own gtest files (values only):
    nocturne/cpp/tests/src/geometry/polygon_test.cc:60-86, intersection_test.cc:52-76."""
    geo = sim_libs.oracle_geo()
    f = lambda pts: np.ascontiguousarray(np.array(pts, np.float32))
    p1 = f([[1, 1], [3, 1], [2, 2]])
    tri = lambda dx, dy: f([[1 + dx, 1 + dy], [3 + dx, 1 + dy], [2 + dx, 2 + dy]])
    assert geo.orageo_poly_poly(p1, 3, tri(0.5, 0.5), 3) == 1
    assert geo.orageo_poly_poly(p1, 3, tri(1.0, 1.0), 3) == 1          # touching corner counts
    assert geo.orageo_poly_poly(p1, 3, tri(5.0, 0.0), 3) == 0
    sq = f([[0, 0], [2, 0], [2, 2], [0, 2]])
    assert geo.orageo_poly_seg(sq, 4, f([1, 1, 3, 3])) == 1
    assert geo.orageo_poly_seg(sq, 4, f([3, 0, 3, 3])) == 0
    assert geo.orageo_poly_seg(sq, 4, f([-1, 3, 3, -1])) == 1
    assert geo.orageo_poly_seg(sq, 4, f([2, 3, 3, 2])) == 0


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rollout_oracle_matches_reference_closed_loop(tag):
    """G8: the unmodified reference AutoregressivePolicy + real FreeCar/Box2D, 20 steps, vs this repo's
    restated loop + C sim: tokens, RTG bins, float32 states and collision flags all identical.  "a", "b" are
    contact-free scenes; in "c" vehicles collide (Box2D's contact solver acts inside the closed loop)."""
    cfg = cfg_of("loop")
    d = spec.Dims(cfg)
    g = golden("closed_loop")
    rc = g[f"{tag}_recipe"]
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]),
                                  n_points=d.NP, extent=float(rc[4]))
    pol = cfg.eval.policy.copy()
    pol.nucleus_sampling = bool(rc[9]); pol.action_temperature = float(rc[10])
    ro = rollout_oracle.RolloutOracle(cfg, weights.generate(d, 0), policy_cfg=pol, tilt=tuple(rc[6:9]), seed=int(rc[5]))
    r = ro.run(scn, 20, sim_libs.OracleSim, record_groups=True)
    if tag == "c":
        assert g[f"{tag}_coll"][..., 0].sum() >= 12             # vehicles collide in this one
    else:
        assert g[f"{tag}_coll"][..., 0].sum() == 0              # contact-free by construction
    assert np.array_equal(r["tokens"], g[f"{tag}_tokens"])
    assert np.array_equal(r["n_groups"], g[f"{tag}_n_groups"])
    np.testing.assert_allclose(fo.undiscretize_rtgs(r["rtg_bins"], cfg.dataset.waymo), g[f"{tag}_rtg_cont"], atol=1e-9)
    np.testing.assert_allclose(r["states"], g[f"{tag}_states"], atol=1e-4, rtol=0)   # north-star tolerance
    assert np.array_equal(r["states"], g[f"{tag}_states"])      # and in fact identical here
    assert np.array_equal(r["coll"], g[f"{tag}_coll"])
    tf = g[f"{tag}_groups_t_focal"]
    assert [(x["t"], x["focal"]) for x in r["groups"]] == [tuple(v) for v in tf]
    for x, ids, mem in zip(r["groups"], g[f"{tag}_groups_ids"], g[f"{tag}_groups_members"]):
        assert x["ids"] == [int(v) for v in ids if v >= 0]
        assert x["members"] == [int(v) for v in mem if v >= 0]
    assert g[f"{tag}_margins"].min() > 1e-4                     # no sampling race was a near-tie


def test_rollout_oracle_matches_reference_closed_loop_at_trained_like_weights():
    """tests/golden/closed_loop_trained.npz "a": the unmodified reference policy + real FreeCar / Box2D with trained-like weights (sharp
    sampling distributions) vs this repo's restated loop: tokens, RTG bins, states, flags identical."""
    cfg = cfg_of("loop")
    d = spec.Dims(cfg)
    g = golden("closed_loop_trained")
    rc = g["a_recipe"]
    steps = int(rc[9])
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP, extent=float(rc[4]))
    ro = rollout_oracle.RolloutOracle(cfg, weights.generate_trained_like(d, 0), tilt=tuple(rc[6:9]), seed=int(rc[5]))
    r = ro.run(scn, steps, sim_libs.OracleSim)
    assert np.array_equal(r["tokens"], g["a_tokens"])
    assert np.array_equal(r["n_groups"], g["a_n_groups"])
    np.testing.assert_allclose(fo.undiscretize_rtgs(r["rtg_bins"], cfg.dataset.waymo), g["a_rtg_cont"], atol=1e-9)
    np.testing.assert_allclose(r["states"], g["a_states"], atol=1e-4, rtol=0)
    assert np.array_equal(r["coll"], g["a_coll"])
    assert len(np.unique(g["a_tokens"])) > 8                                      # not one token repeated


def test_rollout_oracle_matches_reference_on_the_headline_shape():
    """The oracle against tests/golden/closed_loop_wide.npz (64 vehicles x 512 polylines, full model, the unmodified reference
    policy + real physics): the first 3 steps — 14 focal groups and 28 dense forwards per step (the whole 36 steps
    through the window slide are the GPU test's job: the oracle needs ~20 CPU-seconds per step once the window is full)."""
    g = golden("closed_loop_wide")
    rc = g["a_recipe"]
    cfg = spec.make_cfg(nocturne__steps=int(rc[9]))
    d = spec.Dims(cfg)
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP,
                                  extent=float(rc[4]))
    ro = rollout_oracle.RolloutOracle(cfg, weights.generate(d, 0), tilt=tuple(rc[6:9]), seed=int(rc[5]))
    K = 3
    r = ro.run(scn, K, sim_libs.OracleSim, record_groups=True)
    assert np.array_equal(r["tokens"][:, :K], g["a_tokens"][:, :K])
    assert np.array_equal(r["n_groups"][:K], g["a_n_groups"][:K])
    np.testing.assert_allclose(fo.undiscretize_rtgs(r["rtg_bins"][:, :K], cfg.dataset.waymo), g["a_rtg_cont"][:, :K], atol=1e-9)
    assert np.array_equal(r["states"][:, :K + 1], g["a_states"][:, :K + 1])
    assert np.array_equal(r["coll"][:, :K + 1], g["a_coll"][:, :K + 1])
    tf = g["a_groups_t_focal"]
    n0 = int((tf[:, 0] < K).sum())
    assert [(x["t"], x["focal"]) for x in r["groups"]] == [tuple(v) for v in tf[:n0]]
    for x, ids, mem in zip(r["groups"], g["a_groups_ids"][:n0], g["a_groups_members"][:n0]):
        assert x["ids"] == [int(v) for v in ids if v >= 0]
        assert x["members"] == [int(v) for v in mem if v >= 0]


def test_rollout_oracle_matches_reference_on_the_headline_shape_at_trained_like_weights():
    """The oracle against tests/golden/closed_loop_wide_trained.npz (64 vehicles x 512 polylines, full model, trained-like weights, tilts;
    the unmodified reference policy + real physics): the first 2 steps (the whole run is the GPU test's job)."""
    g = golden("closed_loop_wide_trained")
    rc = g["a_recipe"]
    cfg = spec.make_cfg(nocturne__steps=int(rc[9]))
    d = spec.Dims(cfg)
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP, extent=float(rc[4]))
    ro = rollout_oracle.RolloutOracle(cfg, weights.generate_trained_like(d, 0), tilt=tuple(rc[6:9]), seed=int(rc[5]))
    K = 2
    r = ro.run(scn, K, sim_libs.OracleSim)
    assert np.array_equal(r["tokens"][:, :K], g["a_tokens"][:, :K])
    assert np.array_equal(r["n_groups"][:K], g["a_n_groups"][:K])
    np.testing.assert_allclose(fo.undiscretize_rtgs(r["rtg_bins"][:, :K], cfg.dataset.waymo), g["a_rtg_cont"][:, :K], atol=1e-9)
    np.testing.assert_allclose(r["states"][:, :K + 1], g["a_states"][:, :K + 1], atol=1e-4, rtol=0)
    assert np.array_equal(r["coll"][:, :K + 1], g["a_coll"][:, :K + 1])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_rollout_oracle_matches_reference_planner_vs_adversary(tag):
    """Planner-vs-adversary driver (evaluators/planner_adversary_evaluator.py:497-546): two unmodified reference policies
    with the planner / adversary tilts of cfgs/policy/ctrl_sim_{planner,adversary}.yaml, one vehicle each, everybody else
    (and both before history_steps - 1) log-replayed through the reference's inverse bicycle model, real FreeCar/Box2D
    underneath ("b" has the two colliding) — vs the restated loop + C sim: applied actions, RTGs, states, flags identical."""
    import gen_golden
    cfg = cfg_of("loop")
    d = spec.Dims(cfg)
    g = golden("planner_adversary")
    rc = g[f"{tag}_recipe"]
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP,
                                  extent=float(rc[4]))
    ego, adv = gen_golden.pick_ego_adversary(scn)
    assert [ego, adv] == list(g[f"{tag}_ego_adv"])
    ro = rollout_oracle.RolloutOracle(cfg, weights.generate(d, 0), seed=int(rc[5]))
    r = ro.run_planner_adversary(scn, 20, sim_libs.OracleSim, ego, adv, scenarios.standin_log(scn, 20),
                                 int(rc[6]), gen_golden.PLANNER_TILT, gen_golden.ADVERSARY_TILT)
    if tag == "b":
        assert g[f"{tag}_coll"][..., 0].sum() > 0
    np.testing.assert_allclose(r["actions"], g[f"{tag}_actions"], atol=1e-12, rtol=0)
    np.testing.assert_allclose(r["rtg_cont"], g[f"{tag}_rtg_cont"], atol=1e-9, rtol=0)
    assert np.array_equal(r["states"], g[f"{tag}_states"])
    assert np.array_equal(r["coll"], g[f"{tag}_coll"])
    assert g[f"{tag}_margins"].min() > 1e-4


@pytest.mark.parametrize("name,K", [("il", 2), ("trajeglish", 1), ("decision_transformer", 3)])
def test_variant_oracles_match_reference(name, K):
    """cfgs/model/{il,trajeglish}.yaml: token stacks of 2 / 1 types, their causal masks, action logits from token type 0, and
    the single-forward policy — model oracle vs the reference modules' logits, rollout oracle vs the unmodified reference policy
    (tests/golden/variants.npz)."""
    g = golden("variants")
    dt = name == "decision_transformer"
    assert np.array_equal(mo.causal_mask_closed_form(4, 4, K, 1 if dt else 0).numpy(), g[f"{name}_mask_tiny"])
    for tag in ("tiny", "loop"):
        cfg = cfg_of(tag, variant=name)
        d = spec.Dims(cfg)
        assert d.VARIANT == {"il": 1, "trajeglish": 2, "decision_transformer": 3}[name]
        tw = mo.as_torch_weights(weights.generate(d, 0))
        for seed in (1, 2):
            _, t_fill, n_ag, n_pl = [int(v) for v in g[f"{name}_{tag}_s{seed}_recipe"]]
            inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
            if dt:
                inp["rtgs"] = synth_inputs.dt_rtgs(inp["rtgs"], seed)
            with torch.no_grad():
                out = mo.forward(tw, synth_inputs.to_torch(inp), d)
            got = out["action_preds"].numpy() if tag == "tiny" else out["action_preds"][0, :, t_fill - 1].numpy()
            np.testing.assert_allclose(got, g[f"{name}_{tag}_s{seed}_action"], atol=2e-5, rtol=0)
    if dt:
        return                                            # its closed loop needs real-time rewards (separate fixture)
    rc = g[f"{name}_loop_recipe"]
    cfg = cfg_of("loop", variant=name)
    d = spec.Dims(cfg)
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP,
                                  extent=float(rc[4]))
    r = rollout_oracle.RolloutOracle(cfg, weights.generate(d, 0), seed=int(rc[5])).run(scn, 14, sim_libs.OracleSim)
    assert np.array_equal(r["tokens"], g[f"{name}_loop_tokens"]) and np.array_equal(r["n_groups"], g[f"{name}_loop_n_groups"])
    assert np.array_equal(r["states"], g[f"{name}_loop_states"]) and np.array_equal(r["coll"], g[f"{name}_loop_coll"])


def test_own_return_mask_oracles_match_reference():
    """cfg.model.attend_own_return_action = True (cfgs/model/base.yaml:15, utils/train_utils.py:114-129; built in round 6): the closed-form
    mask against the reference's `get_causal_mask`, the model oracle against the reference Encoder / Decoder built with that cfg (both
    heads the policy reads), the rollout oracle against the unmodified reference policy + real FreeCar / Box2D (tests/golden/own_return.npz;
    the same scene rolled under the default mask differs in 7 of 126 tokens: the fixture is not vacuous)."""
    g = golden("own_return")
    assert np.array_equal(mo.causal_mask_closed_form(4, 4, 3, 0, True).numpy(), g["mask_tiny"])
    assert not np.array_equal(mo.causal_mask_closed_form(4, 4, 3, 0, False).numpy(), g["mask_tiny"])
    own = {"model__attend_own_return_action": True}
    from helpers import TINY, LOOP
    for tag, over in (("tiny", TINY), ("loop", LOOP)):
        cfg = spec.make_cfg(**over, **own)
        d = spec.Dims(cfg)
        assert d.MASK_OWN and d.VARIANT == 0
        tw = mo.as_torch_weights(weights.generate(d, 0))
        for seed in (1, 2):
            _, t_fill, n_ag, n_pl = [int(v) for v in g[f"{tag}_s{seed}_recipe"]]
            inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
            with torch.no_grad():
                out = mo.forward(tw, synth_inputs.to_torch(inp), d)
            for head in ("action_preds", "rtg_preds"):
                got = out[head].numpy() if tag == "tiny" else out[head][0, :, t_fill - 1].numpy()
                np.testing.assert_allclose(got, g[f"{tag}_s{seed}_{head}"], atol=2e-5, rtol=0)
    rc = g["loop_recipe"]
    cfg = spec.make_cfg(**LOOP, **own)
    d = spec.Dims(cfg)
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP, extent=float(rc[4]))
    r = rollout_oracle.RolloutOracle(cfg, weights.generate(d, 0), seed=int(rc[5]), tilt=tuple(float(v) for v in rc[6:9])).run(scn, 14, sim_libs.OracleSim)
    assert np.array_equal(r["tokens"], g["loop_tokens"]) and np.array_equal(r["n_groups"], g["loop_n_groups"])
    assert np.array_equal(r["states"], g["loop_states"]) and np.array_equal(r["coll"], g["loop_coll"])
    assert (g["loop_tokens"] != g["loop_tokens_default_mask"]).sum() > 0


FLAG_CASES = {"no_actions": {"model__no_actions": True}, "no_map": {"model__use_map": False},
              "no_init": {"model__encode_initial_state": False}, "no_actions_no_map": {"model__no_actions": True, "model__use_map": False},
                    "own_return_no_init": {"model__attend_own_return_action": True, "model__encode_initial_state": False}}


@pytest.mark.parametrize("name", list(FLAG_CASES))
def test_model_flag_oracles_match_reference(name):
    """cfg.model.no_actions = True / use_map = False / encode_initial_state = False (cfgs/model/base.yaml:4,10; ctrl_sim.yaml:9; built in round 6):
    the model oracle — which follows the reference's branches (modules/encoder.py:129-130,155-170), not the device's key-padding form — against
    the reference Encoder / Decoder built with each cfg, and the rollout oracle against the unmodified reference policy + real FreeCar / Box2D
    (tests/golden/model_flags.npz; the same scene under the shipped cfg gives other tokens)."""
    g = golden("model_flags")
    from helpers import TINY, LOOP
    over = FLAG_CASES[name]
    for tag, dims_over in (("tiny", TINY), ("loop", LOOP)):
        cfg = spec.make_cfg(**dims_over, **over)
        d = spec.Dims(cfg)
        assert d.FLAGS
        tw = mo.as_torch_weights(weights.generate(d, 0))
        for seed in (1, 2):
            _, t_fill, n_ag, n_pl = [int(v) for v in g[f"{name}_{tag}_s{seed}_recipe"]]
            inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
            with torch.no_grad():
                out = mo.forward(tw, synth_inputs.to_torch(inp), d)
            for head in ("action_preds", "rtg_preds"):
                got = out[head].numpy() if tag == "tiny" else out[head][0, :, t_fill - 1].numpy()
                np.testing.assert_allclose(got, g[f"{name}_{tag}_s{seed}_{head}"], atol=2e-5, rtol=0)
    if f"{name}_loop_recipe" not in g.files:
        return
    rc = g[f"{name}_loop_recipe"]
    cfg = spec.make_cfg(**LOOP, **over)
    d = spec.Dims(cfg)
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP, extent=float(rc[4]))
    r = rollout_oracle.RolloutOracle(cfg, weights.generate(d, 0), seed=int(rc[5]), tilt=tuple(float(v) for v in rc[6:9])).run(scn, 14, sim_libs.OracleSim)
    assert np.array_equal(r["tokens"], g[f"{name}_loop_tokens"]) and np.array_equal(r["n_groups"], g[f"{name}_loop_n_groups"])
    assert np.array_equal(r["states"], g[f"{name}_loop_states"]) and np.array_equal(r["coll"], g[f"{name}_loop_coll"])
    assert (g[f"{name}_loop_tokens"] != g[f"{name}_loop_tokens_shipped_cfg"]).sum() > 0


def test_inverse_bicycle_matches_reference():
    """G10: nocturne/bicycle_model.py:51-109 (log-replay actions)."""
    from ctrlsim_amd.kinematics import bicycle_backward
    g = golden("bicycle_backward")
    a, s = bicycle_backward(g["nxt"], g["prev"], 0.1)
    np.testing.assert_allclose(np.stack([a, s], 1), g["accel_steer"], rtol=0, atol=1e-12)


def _kin(state, length, accel, steer, dt, steps=1):
    """Object::KinematicBicycleStep through the oracle (optional integrator mode S6)."""
    import ctypes as C
    lib = C.CDLL(sim_libs.ORA_SO)
    lib.orasim_kinematic_step.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_float]
    st = (C.c_float * 4)(*state)
    for _ in range(steps):
        lib.orasim_kinematic_step(st, length, accel, steer, dt)
    return np.array(st[:], np.float32)


def test_kinematic_step_known_answers_of_the_reference():
    """The known-answer tests the reference holds for Object::KinematicBicycleStep (nocturne/cpp/tests/src/object_test.cc:35-191;
    values transcribed as data), on the oracle's restatement of object.cc:126-137: uniform motion, constant acceleration
    forward and backward, one steering step against the closed form of the test file.  (SpeedCliptTest needs a finite
    max_speed, which the evaluated path never sets: object.h:189.)"""
    q, t, n = np.float32(np.pi / 4), 10.0, 100
    dt = np.float32(t / n)
    # UniformLinearMotionTest: (1,1), heading pi/4, speed 10, 100 steps of 0.1
    st = _kin([1, 1, q, 10], 2.0, 0.0, 0.0, dt, n)
    v = np.float32(10) * np.array([np.cos(q), np.sin(q)], np.float32)
    np.testing.assert_allclose(st[:2], 1 + v * np.float32(t), atol=1e-4)
    assert st[2] == q and st[3] == np.float32(10)
    # ConstantAccelerationMotionTest, forward: speed 0, a = 2
    st = _kin([1, 1, q, 0], 2.0, 2.0, 0.0, dt, n)
    tgt = 1 + np.float32(2) * np.array([np.cos(q), np.sin(q)], np.float32) * np.float32(t * t * 0.5)
    np.testing.assert_allclose(st[:2], tgt, atol=1e-4)
    assert st[2] == q and abs(st[3] - 20.0) < 1e-4
    # backward: speed 10, a = -2
    st = _kin([1, 1, q, 10], 2.0, -2.0, 0.0, dt, n)
    tgt = 1 + v * np.float32(t) - np.float32(2) * np.array([np.cos(q), np.sin(q)], np.float32) * np.float32(t * t * 0.5)
    np.testing.assert_allclose(st[:2], tgt, atol=1e-4)
    assert st[2] == q and abs(st[3] - (-10.0)) < 1e-4
    # SteeringMotionTest: speed 2, steering 10 degrees, one step of 0.1 vs KinematicBicycleModel() of the test file
    f = np.float32
    delta = f(f(10.0) / 180.0 * np.pi)
    beta = f(np.arctan(f(np.tan(delta)) * f(0.5)))
    dx, dy = f(2) * f(np.cos(q + beta)), f(2) * f(np.sin(q + beta))
    dth = f(2) * f(np.tan(delta)) * f(np.cos(beta)) / f(2)
    st = _kin([1, 1, q, 2], 2.0, 0.0, float(delta), 0.1)
    np.testing.assert_allclose(st[:2], [f(1) + dx * f(0.1), f(1) + dy * f(0.1)], rtol=4e-7)      # EXPECT_FLOAT_EQ = 4 ulp
    np.testing.assert_allclose(st[2], q + dth * f(0.1), rtol=4e-7)
    assert st[3] == f(2)


def test_dense_reward_matches_reference_functions():
    """Real-time rewards (ctrlsim_amd/rewards.py: the road-edge signed distance, the nearest-vehicle distance and the reward
    combination of Evaluator.compute_dense_reward) vs the reference's own functions called as the evaluator calls them
    (tests/golden/dense_reward.npz): a single vehicle, an all-but-one-missing scene, cyclic and two-point polylines."""
    from ctrlsim_amd import rewards
    g = golden("dense_reward")
    w = spec.make_cfg().dataset.waymo
    for c in range(4):
        polys = [g[f"c{c}_poly{k}"] for k in range(int(g[f"c{c}_npoly"]))]
        dense, nearest = rewards.dense_reward(g[f"c{c}_xy"], g[f"c{c}_exist"], g[f"c{c}_rewards"][:, 0], polys, w)
        np.testing.assert_allclose(dense, g[f"c{c}_dense"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(nearest * w.max_veh_veh_distance, g[f"c{c}_nearest_metric"], rtol=0, atol=1e-12)
        sd = -rewards.signed_distance_to_road_edges(g[f"c{c}_xy"], polys) / w.dist_to_road_edge_scaling_factor * g[f"c{c}_exist"]
        np.testing.assert_allclose(sd, g[f"c{c}_edge_signed"], rtol=0, atol=1e-12)


def test_rollout_oracle_matches_reference_decision_transformer_loop():
    """cfgs/policy/dt.yaml (real_time_rewards, max_return, continuous RTGs) in closed loop: the unmodified reference policy and
    model + the reference's reward functions + real FreeCar/Box2D (tests/golden/dt_loop.npz) vs the restated loop."""
    g = golden("dt_loop")
    rc = g["loop_recipe"]
    cfg = cfg_of("loop", variant="decision_transformer")
    d = spec.Dims(cfg)
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP,
                                  extent=float(rc[4]))
    r = rollout_oracle.RolloutOracle(cfg, weights.generate(d, 0), seed=int(rc[5])).run(scn, 14, sim_libs.OracleSim)
    assert np.array_equal(r["tokens"], g["loop_tokens"])
    np.testing.assert_allclose(r["rtgs"], g["loop_rtgs"], rtol=0, atol=1e-9)
    assert np.array_equal(r["states"], g["loop_states"]) and np.array_equal(r["coll"], g["loop_coll"])
    assert g["loop_rtgs"][:, -1, 1].max() < 90.0               # the vehicle-distance RTG is being spent


@pytest.mark.skipif(not sim_libs.RefSim.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_general_set_position_matches_real_box2d_live():
    """Vehicle::set_position at an ARBITRARY target (vehicle.cc:75-87 -> BaseCar::SetPosition = b2Body::SetTransform at the current
    angle, physics/BaseCar.cpp:28-32) — not only the (-1e6, -1e6) parking the rollout uses: a vehicle is dropped onto another one
    mid-run, a second one moved away and back.  The C oracle against the real FreeCar + Box2D, bit-exact, through the contacts
    that follow."""
    scn = scenarios.make_scenario(5, 1, n_agents=6, n_polylines=12, n_points=10, extent=30.0)
    sims = [cls(scn.length, scn.width, scn.x, scn.y, scn.heading, scn.speed, scn.edge_segments) for cls in (sim_libs.RefSim, sim_libs.OracleSim)]
    rs = np.random.RandomState(3)
    hit = 0
    for t in range(16):
        acts = np.stack([rs.uniform(-3, 3, 6), rs.uniform(-0.3, 0.3, 6)], 1)
        st0 = sims[0].state()[0]
        for sm in sims:
            if t == 5:
                sm.set_position(2, float(st0[0, 0]) + 0.5, float(st0[0, 1]) + 0.25)      # onto vehicle 0
            if t == 8:
                sm.set_position(4, 500.0, -300.0)
            if t == 11:
                sm.set_position(4, float(st0[1, 0]) - 1.0, float(st0[1, 1]))             # back, onto vehicle 1
            for i in range(6):
                sm.set_action(i, float(acts[i, 0]), float(acts[i, 1]))
            sm.step(0.1)
        (a, av, ae), (b, bv, be) = sims[0].state(), sims[1].state()
        assert np.array_equal(a, b), t
        assert np.array_equal(av, bv) and np.array_equal(ae, be), t
        hit += int(av.sum())
    assert hit > 0
    for sm in sims:
        sm.close()


def test_oracle_expert_controlled_vehicles_match_real_box2d_live():
    """Scenario::Step with expert-controlled objects (nocturne/cpp/src/scenario.cc:272-284): the physics step moves every body, then an
    expert vehicle is put on the LOGGED position, heading and speed through Vehicle::set_position / set_heading / set_speed
    (vehicle.cc:75-105: two b2Body::SetTransform — proxy synchronisation, new-contact search at the top of the next step — and
    SetLinearVelocity, which wakes the body).  Two of six vehicles follow a log that drives one of them THROUGH a policy vehicle, a third
    switches between expert and free control mid-run, one expert stands still (zero velocity: not woken).  The C oracle against the real
    FreeCar + Box2D, bit for bit, collision flags included."""
    scn = scenarios.make_scenario(11, 2, n_agents=6, n_polylines=12, n_points=10, extent=30.0)
    sims = [cls(scn.length, scn.width, scn.x, scn.y, scn.heading, scn.speed, scn.edge_segments) for cls in (sim_libs.RefSim, sim_libs.OracleSim)]
    rs = np.random.RandomState(7)
    T = 24
    # logs: vehicle 1 drives a straight line through vehicle 0's start position; vehicle 3 circles; vehicle 5 stands still
    tt = np.arange(1, T + 1, dtype=np.float32)
    log = {1: np.stack([scn.x[0] - 6.0 + 0.6 * tt, np.full(T, scn.y[0] + 0.3, np.float32), np.full(T, 0.05, np.float32), np.full(T, 6.0, np.float32)], 1),
           3: np.stack([scn.x[3] + 4 * np.cos(0.2 * tt), scn.y[3] + 4 * np.sin(0.2 * tt), 0.2 * tt + np.float32(np.pi / 2), np.full(T, 0.8, np.float32)], 1),
           5: np.stack([np.full(T, scn.x[5]), np.full(T, scn.y[5]), np.full(T, scn.heading[5]), np.zeros(T, np.float32)], 1)}
    hit = 0
    for t in range(T):
        acts = np.stack([rs.uniform(-3, 3, 6), rs.uniform(-0.3, 0.3, 6)], 1)
        st0 = sims[0].state()[0]
        if 5 <= t < 12:                                         # the log of vehicle 1 runs over vehicle 0 for a while
            log[1][t, 0], log[1][t, 1] = st0[0, 0] + np.float32(1.5 - 0.4 * (t - 5)), st0[0, 1] + np.float32(0.3)
        for sm in sims:
            for i in range(6):
                sm.set_action(i, float(acts[i, 0]), float(acts[i, 1]))
            for i, lg in log.items():
                if i == 3 and 8 <= t < 14:
                    continue                                    # free control for a while, then expert again
                sm.set_expert(i, float(lg[t, 0]), float(lg[t, 1]), float(lg[t, 2]), float(lg[t, 3]))
            sm.step(0.1)
        (a, av, ae), (b, bv, be) = sims[0].state(), sims[1].state()
        assert np.array_equal(a, b), t
        assert np.array_equal(av, bv) and np.array_equal(ae, be), t
        assert np.array_equal(sims[0].body(), sims[1].body()), t
        if not (8 <= t < 14):
            assert np.array_equal(a[3, :4], log[3][t].astype(np.float32)), t      # an expert vehicle reads back its log
        hit += int(av.sum())
    assert hit > 0
    for sm in sims:
        sm.close()
