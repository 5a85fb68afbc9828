"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by RUNNING THE REFERENCE ITSELF in the build
container (needs /root/reference and oracle/_ref/libref_sim.so).  Fixtures are data only: recipes
(dims + seeds) as inputs, reference outputs as arrays.  Usage:  python oracle/gen_golden.py [names...]

  G1 mask        closed form == utils/train_utils.py:get_causal_mask          (asserted here)
  G2 model_tiny  reference Encoder+Decoder, A=4,T=4,P=6,NP=8: all three heads, full
  G3 model_full  reference Encoder+Decoder, A=24,T=32,P=200,NP=100: logits at token_index only
  G4 features    reference select_relevant_agents / discretize_* / normalize_scene via get_data()
  G5 sampling    reference process_predicted_rtg / action sampling code path with patched multinomial
  G6 physics     real FreeCar + Box2D trajectories under scripted actions (+ a contact case, informational)
  G7 collision   real ConvexPolygon::Intersects / polygon-segment on random boxes + the reference's test KATs
  G8 closed_loop reference AutoregressivePolicy.update_state/predict/act + real physics, 8 agents x 20 steps
  G9 kinematic   KATs of nocturne/cpp/tests/src/object_test.cc (values transcribed as data)
  G10 bicycle    nocturne/bicycle_model.py BicycleModel.backward on random pairs
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ctrlsim_amd  # noqa: E402
from ctrlsim_amd import spec, weights, scenarios  # noqa: E402
import ref_shims  # noqa: E402
import synth_inputs  # noqa: E402
import model_oracle  # noqa: E402
from sim_libs import RefSim, OracleSim, ref_geo  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TINY = dict(dataset__waymo__max_num_agents=4, dataset__waymo__train_context_length=4,
            dataset__waymo__max_num_road_polylines=6, dataset__waymo__max_num_road_pts_per_polyline=8)
# closed-loop config: small context so that >1 focal group and the P_all>P selection are exercised cheaply
LOOP = dict(dataset__waymo__max_num_agents=6, dataset__waymo__train_context_length=8,
            dataset__waymo__max_num_road_polylines=12, dataset__waymo__max_num_road_pts_per_polyline=10,
            nocturne__steps=20)


def save(name, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KB")


# --------------------------------------------------------------------------------------------- G1-G3
def gen_model():
    for tag, over in (("tiny", TINY), ("full", {})):
        cfg = spec.make_cfg(**over)
        d = spec.Dims(cfg)
        w = weights.generate(d, 0)
        ref = ref_shims.build_reference_model(cfg, w)
        cm = model_oracle.causal_mask_closed_form(d.A, d.T, 3)
        assert bool(((ref.decoder.causal_mask == 0) == cm).all()), "closed-form mask != get_causal_mask"
        out = {"mask_visible_fraction": np.float64(cm.float().mean().item())}
        cases = [(1, d.T, d.A - 1, d.P - 2), (2, max(1, d.T // 2), d.A, d.P)] if tag == "tiny" else \
                [(1, d.T, d.A - 3, d.P - 10), (2, 6, d.A, d.P)]
        for seed, t_fill, n_ag, n_pl in cases:
            inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
            r = ref(synth_inputs.to_motion_data(inp), eval=True)      # autograd ON, like the reference
            ti = t_fill - 1
            if tag == "tiny":
                for k in ("action_preds", "rtg_preds", "state_preds"):
                    out[f"s{seed}_{k}"] = r[k].detach().numpy()
            else:
                out[f"s{seed}_action_logits"] = r["action_preds"][0, :, ti].detach().numpy()
                out[f"s{seed}_rtg_logits"] = r["rtg_preds"][0, :, ti].detach().numpy()
            out[f"s{seed}_recipe"] = np.array([seed, t_fill, n_ag, n_pl])
        save(f"model_{tag}", **out)


def gen_model_trained():
    """Round 5: the unmodified reference Encoder / Decoder at FULL dims with trained-like weights (weights.generate_trained_like:
    LayerNorm gains in [0.5, 2], per-row / per-column matrix scales, embedding rows over three decades, query / key rows x 2, head
    gain 15: |logit| up to ~30, max probabilities 0.3-0.96).  Next to the float32 reference logits the fixture holds the float64
    evaluation of the same network (the oracle's functional form in double) so that tests can see how far float32 itself is from
    exact arithmetic in this regime."""
    cfg = spec.make_cfg()
    d = spec.Dims(cfg)
    out = {}
    for wseed, (seed, t_fill, n_ag, n_pl) in ((0, (1, d.T, d.A - 3, d.P - 10)), (1, (2, 9, d.A, d.P)), (0, (3, d.T, 7, d.P - 40))):
        w = weights.generate_trained_like(d, wseed)
        ref = ref_shims.build_reference_model(cfg, w)
        inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=n_ag, n_polys=n_pl)
        r = ref(synth_inputs.to_motion_data(inp), eval=True)
        ti = t_fill - 1
        with torch.no_grad():
            orig = torch.Tensor.float
            torch.Tensor.float = lambda self, *a, **k: self.double()
            try:
                o64 = model_oracle.forward({k: torch.from_numpy(v).double() for k, v in w.items()}, synth_inputs.to_torch(inp), d)
            finally:
                torch.Tensor.float = orig
        for k, nm in (("action_preds", "action_logits"), ("rtg_preds", "rtg_logits")):
            out[f"s{seed}_{nm}"] = r[k][0, :, ti].detach().numpy()
            out[f"s{seed}_{nm}_f64"] = o64[k][0, :n_ag, ti].numpy()          # live slots only
            e = np.abs(out[f"s{seed}_{nm}"][:n_ag] - out[f"s{seed}_{nm}_f64"]).max()
            print(seed, nm, "max|logit|", np.abs(out[f"s{seed}_{nm}"][:n_ag]).max(), "reference fp32 vs float64", e)
        out[f"s{seed}_recipe"] = np.array([seed, t_fill, n_ag, n_pl, wseed])
    save("model_trained", **out)


def gen_closed_loop_trained():
    """Closed loops of the unmodified reference policy + real FreeCar / Box2D at the trained-like weights: "a" the small closed-loop
    model (LOOP dims, 10 vehicles x 20 steps, sharp distributions: the sampled ids are mostly the arg max), "b" FULL dims, 10 vehicles
    x 230 polylines x 34 steps (two past the window slide; the reference sizes its window by min(steps, T), so a full-dims loop needs
    steps >= T)."""
    out = {}
    for tag, over, n_ag, n_pl, steps, extent, tilt in (("a", LOOP, 10, 14, 20, 40.0, (0.0, 0.0, 0.0)),
                                                        ("b", dict(nocturne__steps=34), 10, 230, 34, 40.0, (5.0, -10.0, 10.0))):
        cfg = spec.make_cfg(**over)
        d = spec.Dims(cfg)
        w = weights.generate_trained_like(d, 0)
        scn = scenarios.make_scenario(13, {"a": 0, "b": 1}[tag], n_agents=n_ag, n_polylines=n_pl, n_points=d.NP, extent=extent)
        r = ref_closed_loop(cfg, w, scn, steps, seed=6, tilt=tilt)
        print(tag, "groups/step", r["n_groups"], "min race margin", r["margins"].min(), "collisions", r["coll"].sum(0).sum(0),
              "distinct tokens", len(np.unique(r["tokens"])))
        for k in ("tokens", "rtg_cont", "states", "coll", "actions", "n_groups", "margins"):
            out[f"{tag}_{k}"] = r[k]
        out[f"{tag}_recipe"] = np.array([13, {"a": 0, "b": 1}[tag], n_ag, n_pl, extent, 6, *tilt, steps])
    save("closed_loop_trained", **out)


# --------------------------------------------------------------------------------------------- reference closed loop
class _FakeVeh:
    """The slice of the pybind Vehicle surface that Policy.act touches (pybind11/src/object.cc:33-99)."""

    def __init__(self, sim, i):
        self._sim, self._i = sim, i
        self._accel, self._steer = 0.0, 0.0
        self.calls = []

    def getID(self):
        return self._i

    def setPosition(self, x, y):
        self._sim.set_position(self._i, x, y)

    @property
    def acceleration(self):
        return self._accel

    @acceleration.setter
    def acceleration(self, v):
        self._pending = ("throttle", float(v))

    def brake(self, v):
        self._pending = ("brake", float(v))

    @property
    def steering(self):
        return self._steer

    @steering.setter
    def steering(self, v):
        kind, val = self._pending
        # replay through the harness entry point, which performs the same Throttle/Brake/Turn calls
        self._sim.set_action(self._i, val if kind == "throttle" else -val, float(v))


class _NoisePatch:
    """torch.multinomial(p,1) -> argmax(p/q) with q keyed by (seed, scenario, t, veh, head); the calling
    frame tells which vehicle/head is being sampled (policy.py:123-127, autoregressive_policy.py:236)."""

    def __init__(self, seed, scenario):
        self.seed, self.scenario, self.t = seed, scenario, 0
        self.count = {}
        self.log = []

    def __call__(self, probs, n):
        fr = sys._getframe(1)
        veh = int(fr.f_locals["veh_id"])
        if fr.f_code.co_name == "process_predicted_rtg":
            head = self.count.get((self.t, veh), 0)
            self.count[(self.t, veh)] = head + 1
            assert head < 3
        else:
            head = 3
        q = weights.exp_noise(self.seed, self.scenario, self.t, veh, head, probs.shape[0])
        ratio = probs / torch.from_numpy(q).to(probs.dtype)
        top2 = torch.topk(ratio, 2).values
        self.log.append(float((top2[0] - top2[1]) / top2[0]))       # relative margin of the race
        return torch.argmax(ratio).reshape(1)


def ref_closed_loop(cfg, w, scn, steps, seed, tilt=(0, 0, 0), nucleus=False, temperature=1.0, rtgs=True):
    """evaluate_policy's inner loop (policy_evaluator.py:514-557) around the UNMODIFIED reference policy.
    rtgs=False: the IL / Trajeglish policies (cfgs/policy/{il,trajeglish}.yaml: use_rtg = predict_rtgs = False)."""
    ref_shims.install()
    from policies.autoregressive_policy import AutoregressivePolicy

    model = ref_shims.build_reference_model(cfg, variant_weights(cfg, w))
    dset = ref_shims.build_reference_dataset(cfg)
    key_dict = {"next_acceleration": "next_acceleration", "next_steering": "next_steering", "rtgs": "rtgs"}
    tilt_dict = {"tilt": True, "goal_tilt": tilt[0], "veh_veh_tilt": tilt[1], "veh_edge_tilt": tilt[2]}
    pol = AutoregressivePolicy(cfg=cfg, model_path="", model=model, use_rtg=rtgs, predict_rtgs=rtgs,
                               discretize_rtgs=True, real_time_rewards=False, privileged_return=False,
                               max_return=False, min_return=False, key_dict=key_dict, tilt_dict=tilt_dict,
                               name="ctrl_sim", action_temperature=temperature, nucleus_sampling=nucleus,
                               nucleus_threshold=0.8)
    N = scn.N
    sim = RefSim(scn.length, scn.width, scn.x, scn.y, scn.heading, scn.speed, scn.edge_segments)
    vehs = [_FakeVeh(sim, i) for i in range(N)]
    vdd = {}
    for i in range(N):
        vdd[i] = {"position": [], "velocity": [], "heading": [], "existence": [], "acceleration": [],
                  "steering": [], "timestep": [], "rtgs": [], "next_acceleration": 0., "next_steering": 0.,
                  "goal_position": {"x": scn.goal_pos[i, 0], "y": scn.goal_pos[i, 1]},
                  "goal_heading": scn.goal_heading[i], "goal_speed": scn.goal_speed[i],
                  "width": scn.width[i], "length": scn.length[i], "type": "vehicle"}
    gt = {i: {"traj": np.ones((91, 6))} for i in range(N)}
    preproc = {"road_points": scn.road_points.astype(np.float64), "road_types": scn.road_types.copy()}
    to_eval = list(range(N))
    patch = _NoisePatch(seed, scn.index)
    orig = torch.multinomial
    torch.multinomial = patch
    states = np.zeros((N, steps + 1, 8))
    coll = np.zeros((N, steps + 1, 2), np.uint8)
    tokens = np.zeros((N, steps), np.int64)
    rtg_cont = np.zeros((N, steps, 3))
    applied = np.zeros((N, steps, 2))
    n_groups = np.zeros(steps, np.int64)
    groups_log = []

    def update(t):
        st, cv, ce = sim.state()
        for i in range(N):
            vdd[i]["position"].append({"x": st[i, 0], "y": st[i, 1]})
            vdd[i]["velocity"].append({"x": st[i, 4], "y": st[i, 5]})
            vdd[i]["heading"].append(st[i, 2])
            vdd[i]["timestep"].append(t)
            vdd[i]["existence"].append(1.0)
            states[i, t] = [st[i, 0], st[i, 1], st[i, 4], st[i, 5], st[i, 2], scn.length[i], scn.width[i], 1.0]
        coll[:, t, 0], coll[:, t, 1] = cv, ce

    try:
        pol.reset(vdd)
        orig_get_data = pol.get_data

        def spy_get_data(*a, **k):
            md, dead, idx_dicts, veh_ids = orig_get_data(*a, **k)
            for focal in md.keys():
                groups_log.append((patch.t, int(focal), sorted(int(x) for x in idx_dicts[focal].keys()),
                                   [int(x) for x in veh_ids[focal]]))
            n_groups[patch.t] = len(md)
            return md, dead, idx_dicts, veh_ids

        pol.get_data = spy_get_data
        for t in range(steps):
            patch.t = t
            update(t)
            pol.update_state(vdd, to_eval, t)
            vdd = pol.predict(vdd, gt, preproc, dset, to_eval, t)
            for i in range(N):
                _, act = pol.act(vehs[i], t, vdd)
                vdd[i]["acceleration"].append(act[0])
                vdd[i]["steering"].append(act[1])
                applied[i, t] = act
                if rtgs:
                    rtg_cont[i, t] = vdd[i]["rtgs"][-1]
            tokens[:, t] = dset.discretize_actions(applied[:, t:t + 1].copy())[:, 0]
            sim.step(0.1)
        update(steps)
    finally:
        torch.multinomial = orig
        sim.close()
    return dict(tokens=tokens, rtg_cont=rtg_cont, states=states, coll=coll, actions=applied, n_groups=n_groups,
                margins=np.array(patch.log), groups=groups_log)


def gen_closed_loop():
    cfg = spec.make_cfg(**LOOP)
    d = spec.Dims(cfg)
    w = weights.generate(d, 0)
    out = {}
    for tag, n_ag, n_pl, tilt, nucleus, temp in (("a", 8, 20, (0, 0, 0), False, 1.0),
                                                  ("b", 10, 9, (5.0, -10.0, 20.0), True, 1.5),
                                                  ("c", 10, 12, (0.0, -10.0, 0.0), False, 1.0)):
        # "a", "b": the first scenario index whose reference rollout never overlaps two car boxes (contact-free);
        # "c": the first one in which vehicles DO collide (Box2D's contact solver in the loop), vehicle-vehicle tilt -10
        for idx in range({"a": 0, "b": 100, "c": 200}[tag], 1000):
            scn = scenarios.make_scenario(7, idx, n_agents=n_ag, n_polylines=n_pl, n_points=d.NP, extent={"c": 22.0}.get(tag, 40.0))
            r = ref_closed_loop(cfg, w, scn, 20, seed=3, tilt=tilt, nucleus=nucleus, temperature=temp)
            print(tag, idx, "veh-veh flags", r["coll"][..., 0].sum())
            if (r["coll"][..., 0].sum() == 0) if tag != "c" else (r["coll"][..., 0].sum() >= 12):
                break
        print(tag, "groups/step", r["n_groups"], "min race margin", r["margins"].min(),
              "collisions", r["coll"].sum(0).sum(0))
        for k in ("tokens", "rtg_cont", "states", "coll", "actions", "n_groups", "margins"):
            out[f"{tag}_{k}"] = r[k]
        g = r["groups"]
        out[f"{tag}_groups_t_focal"] = np.array([(t, f) for t, f, _, _ in g])
        out[f"{tag}_groups_ids"] = np.array([ids + [-1] * (d.A - len(ids)) for _, _, ids, _ in g])
        out[f"{tag}_groups_members"] = np.array([m + [-1] * (n_ag - len(m)) for _, _, _, m in g])
        out[f"{tag}_recipe"] = np.array([7, idx, n_ag, n_pl, {"c": 22.0}.get(tag, 40.0), 3, *tilt, int(nucleus), temp])
    save("closed_loop", **out)


def gen_closed_loop_full():
    """The unmodified reference policy + real FreeCar/Box2D at the FULL model dims (A=24, T=32, P=200, 100 points) through
    the sliding-window phase: 12 vehicles x 40 steps over 230 polylines (P_all > P: nearest-polyline selection every
    step).  Steps 32..39 re-origin the frame at the focal agent's pose of window index 0 = t-31
    (autoregressive_policy.py:55-70, dataset.py:390-394).  Second case: 30 vehicles (> 24 context slots: several focal
    groups per step) x 36 steps."""
    out = {}
    for tag, n_ag, n_pl, steps, extent, tilt in (("a", 12, 230, 40, 40.0, (0.0, 0.0, 0.0)),
                                                  ("b", 30, 210, 36, 45.0, (5.0, -10.0, 10.0))):
        cfg = spec.make_cfg(nocturne__steps=steps)
        d = spec.Dims(cfg)
        w = weights.generate(d, 0)
        scn = scenarios.make_scenario(11, {"a": 0, "b": 1}[tag], n_agents=n_ag, n_polylines=n_pl, n_points=d.NP, extent=extent)
        import time
        t0 = time.time()
        r = ref_closed_loop(cfg, w, scn, steps, seed=5, tilt=tilt)
        print(tag, f"{time.time() - t0:.0f} s; groups/step", r["n_groups"], "min race margin", r["margins"].min(),
              "collisions", r["coll"].sum(0).sum(0))
        for k in ("tokens", "rtg_cont", "states", "coll", "actions", "n_groups", "margins"):
            out[f"{tag}_{k}"] = r[k]
        g = r["groups"]
        out[f"{tag}_groups_t_focal"] = np.array([(t, f) for t, f, _, _ in g])
        out[f"{tag}_groups_ids"] = np.array([ids + [-1] * (d.A - len(ids)) for _, _, ids, _ in g])
        out[f"{tag}_groups_members"] = np.array([m + [-1] * (n_ag - len(m)) for _, _, _, m in g])
        out[f"{tag}_recipe"] = np.array([11, {"a": 0, "b": 1}[tag], n_ag, n_pl, extent, 5, *tilt, steps])
    save("closed_loop_full", **out)



def gen_closed_loop_wide_trained():
    """Round 5: a second scene of the HEADLINE shape (64 vehicles x 512 polylines, full model) rolled by the unmodified reference policy + real
    FreeCar / Box2D, this time with TRAINED-LIKE weights (weights.generate_trained_like: sharp sampling distributions) and tilted RTG
    sampling, 34 steps (two past the window slide): ~14 focal groups per step, 8 704 more sampled ids of the headline shape against the reference
    itself, in the regime a trained checkpoint runs in."""
    import time
    steps = 34
    cfg = spec.make_cfg(nocturne__steps=steps)
    d = spec.Dims(cfg)
    w = weights.generate_trained_like(d, 0)
    seed0, idx, n_ag, n_pl, extent, seed, tilt = 0, 11, 64, 512, 100.0, 4, (5.0, -10.0, 10.0)
    scn = scenarios.make_scenario(seed0, idx, n_agents=n_ag, n_polylines=n_pl, n_points=d.NP, extent=extent)
    t0 = time.time()
    r = ref_closed_loop(cfg, w, scn, steps, seed=seed, tilt=tilt)
    print(f"{time.time() - t0:.0f} s; groups/step", r["n_groups"], "min race margin", r["margins"].min(),
          "collisions", r["coll"].sum(0).sum(0), "distinct tokens", len(np.unique(r["tokens"])))
    out = {}
    for k in ("tokens", "rtg_cont", "states", "coll", "actions", "n_groups", "margins"):
        out[f"a_{k}"] = r[k]
    out["a_recipe"] = np.array([seed0, idx, n_ag, n_pl, extent, seed, *tilt, steps])
    save("closed_loop_wide_trained", **out)


def gen_closed_loop_wide():
    """The HEADLINE shape (BASELINE configs[2]) against the reference itself: ONE scene of 64 vehicles x 512 polylines (the bench's
    generator and extent), full model dims, unmodified reference policy + real FreeCar/Box2D for 36 steps — ~14 focal groups
    per step, the nearest-200-of-512 polyline selection, vehicles dropped from 24-slot contexts, 4 steps past the window slide
    (autoregressive_policy.py:55-70,96-163, datasets/rl_waymo/dataset.py:278-319,390-428)."""
    import time
    steps = 36
    cfg = spec.make_cfg(nocturne__steps=steps)
    d = spec.Dims(cfg)
    w = weights.generate(d, 0)
    seed0, idx, n_ag, n_pl, extent, seed, tilt = 0, 5, 64, 512, 100.0, 9, (0.0, 0.0, 0.0)
    scn = scenarios.make_scenario(seed0, idx, n_agents=n_ag, n_polylines=n_pl, n_points=d.NP, extent=extent)
    t0 = time.time()
    r = ref_closed_loop(cfg, w, scn, steps, seed=seed, tilt=tilt)
    print(f"{time.time() - t0:.0f} s; groups/step", r["n_groups"], "min race margin", r["margins"].min(),
          "collisions", r["coll"].sum(0).sum(0))
    out = {}
    for k in ("tokens", "rtg_cont", "states", "coll", "actions", "n_groups", "margins"):
        out[f"a_{k}"] = r[k]
    g = r["groups"]
    out["a_groups_t_focal"] = np.array([(t, f) for t, f, _, _ in g])
    out["a_groups_ids"] = np.array([ids + [-1] * (d.A - len(ids)) for _, _, ids, _ in g])
    out["a_groups_members"] = np.array([m + [-1] * (n_ag - len(m)) for _, _, _, m in g])
    out["a_recipe"] = np.array([seed0, idx, n_ag, n_pl, extent, seed, *tilt, steps])
    save("closed_loop_wide", **out)


# --------------------------------------------------------------------------------------------- E2: evaluator metrics
def _install_evaluator_stubs():
    """Make the reference's evaluators/policy_evaluator.py importable: it pulls in nocturne (pybind, unbuildable here), imageio,
    matplotlib, hydra and the dataset package at module top.  Only the pybind surface is faked (an enum and a class name); the
    arithmetic under test — Evaluator.initialize_goal_dict / compute_goal_dist_normalizer / compute_nearest_dist_all,
    PolicyEvaluator.update_vehicle_data_dict / update_running_statistics / compute_metrics, utils.sim.compute_reward,
    RLWaymoDataset.compute_dist_to_nearest_vehicle_rewards, scipy's jensenshannon — is the reference's own code."""
    import types
    ref_shims.install()
    if "evaluators.policy_evaluator" in sys.modules:
        return
    REF = ref_shims.REF

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class CollisionType:
        UNCOLLIDED, VEHICLE_VEHICLE, VEHICLE_ROAD = 0, 1, 2
    noct = stub("nocturne", Simulation=object, CollisionType=CollisionType)
    noct.__path__ = [f"{REF}/nocturne"]                  # nocturne.bicycle_model is plain Python
    for name in ("imageio", "matplotlib", "matplotlib.pyplot", "pdb"):
        try:
            __import__(name)
        except Exception:
            stub(name)
    cfgmod = sys.modules["cfgs.config"]
    cfgmod.set_display_window = lambda: None
    cfgmod.get_scenario_dict = lambda cfg: {}
    stub("utils.viz")
    rw = sys.modules["datasets.rl_waymo"]
    from datasets.rl_waymo.dataset import RLWaymoDataset
    rw.RLWaymoDatasetCtRLSim = RLWaymoDataset
    rw.RLWaymoDatasetCTGPlusPlus = type("RLWaymoDatasetCTGPlusPlus", (), {})
    pol = sys.modules["policies"]
    from policies.autoregressive_policy import AutoregressivePolicy
    pol.AutoregressivePolicy = AutoregressivePolicy
    pol.CTGPlusPlusPolicy = type("CTGPlusPlusPolicy", (), {})
    ev = types.ModuleType("evaluators")
    ev.__path__ = [f"{REF}/evaluators"]
    sys.modules["evaluators"] = ev


class _XY:
    def __init__(self, x, y):
        self.x, self.y = x, y


class _ReplayVeh:
    """The read side of the pybind Vehicle (pybind11/src/object.cc:33-99) replaying recorded states."""

    def __init__(self, i, scn, states, coll):
        self.i, self.scn, self.states, self.coll, self.t = i, scn, states, coll, 0
        self.target_position = _XY(float(scn.goal_pos[i, 0]), float(scn.goal_pos[i, 1]))
        self.target_heading, self.target_speed = float(scn.goal_heading[i]), float(scn.goal_speed[i])

    def getID(self): return self.i
    def getWidth(self): return float(self.scn.width[self.i])
    def getLength(self): return float(self.scn.length[self.i])
    def getPosition(self): return _XY(self.states[self.i, self.t, 0], self.states[self.i, self.t, 1])
    def velocity(self): return _XY(self.states[self.i, self.t, 2], self.states[self.i, self.t, 3])
    def getHeading(self): return self.states[self.i, self.t, 4]
    @property
    def position(self): return self.getPosition()
    @property
    def speed(self): return float(np.hypot(self.states[self.i, self.t, 2], self.states[self.i, self.t, 3]))
    @property
    def heading(self): return self.getHeading()
    @property
    def collision_type_veh(self): return 1 if self.coll[self.i, self.t, 0] else 0
    @property
    def collision_type_edge(self): return 2 if self.coll[self.i, self.t, 1] else 0


def metrics_cases():
    """(tag, closed-loop fixture tag, history_steps, vehicles that leave the log at step (veh, t), evaluated vehicles)."""
    return (("a", "a", 1, (), (0, 1, 2, 3, 4, 5, 6, 7)), ("b", "b", 5, ((1, 12), (3, 3), (7, 19)), (0, 1, 3, 4, 7, 9)),
            ("c", "c", 3, ((0, 8),), (0, 2, 5, 6, 8)))


def gen_metrics():
    """The reference's evaluator bookkeeping on recorded rollouts (the reference closed loops of closed_loop.npz): per-step
    rewards / nearest distances (update_vehicle_data_dict -> compute_reward, compute_nearest_dist_all), goal relocation for
    vehicles that leave the log (initialize_goal_dict), the running statistics and the 9 metrics incl. scipy's Jensen-Shannon
    distance (policy_evaluator.py:162-305)."""
    import types as _t
    _install_evaluator_stubs()
    from evaluators.policy_evaluator import PolicyEvaluator
    from ctrlsim_amd.scenarios import standin_log
    g = np.load(os.path.join(GOLD, "closed_loop.npz"))
    out = {}
    base = spec.make_cfg(**LOOP)
    ev = PolicyEvaluator.__new__(PolicyEvaluator)
    ev.cfg, ev.cfg_rl_waymo = base, base.dataset.waymo
    ev.steps, ev.dt = base.nocturne.steps, base.nocturne.dt
    ev.policy = _t.SimpleNamespace(real_time_rewards=False)
    ev.preprocessed_dset = ref_shims.build_reference_dataset(base)
    ev.reset()
    d = spec.Dims(base)
    for tag, src, hist, leave, evals in metrics_cases():
        rc = g[f"{src}_recipe"]
        scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP,
                                      extent=float(rc[4]))
        states, coll, actions = g[f"{src}_states"], g[f"{src}_coll"], g[f"{src}_actions"]
        N, T1 = states.shape[:2]
        gt = standin_log(scn, ev.steps, ev.dt)
        for v, t0 in leave:
            gt[v]["traj"][t0:, 4] = 0.0
        ev.history_steps = hist
        ev.vehicles_to_evaluate = list(evals)
        vehs = [_ReplayVeh(i, scn, states, coll) for i in range(N)]
        vdd, goal_dict, goal_norm = {}, {}, {}
        for veh in vehs:
            v = veh.getID()
            goal_dict[v] = ev.initialize_goal_dict(veh, np.array(gt[v]["traj"]))
            # the list fields of initialize_vehicle_data_dict (policy_evaluator.py:70-96; its scalar fields are not read here)
            vdd[v] = {k: [] for k in ("gt_position", "gt_speed", "gt_heading", "gt_acceleration", "gt_nearest_dist", "position",
                                      "velocity", "heading", "nearest_dist", "existence", "acceleration", "steering", "reward",
                                      "dense_reward", "timestep", "rtgs")}
            goal_norm[v] = ev.compute_goal_dist_normalizer(veh, goal_dict[v]["pos"])
        for t in range(T1):
            for veh in vehs:
                veh.t = t
            vdd = ev.update_vehicle_data_dict(t, vehs, vdd, goal_dict, goal_norm, gt, None, None)
            for v in range(N):
                vdd[v]["acceleration"].append(actions[v, t, 0] if t < ev.steps else 0)
        n0 = [len(x) for x in (ev.goal_achieved_all, ev.collision_rate_scenario, ev.ades_all)]
        ev.update_running_statistics(vdd)
        out[f"{tag}_gt"] = np.stack([np.array(gt[v]["traj"]) for v in range(N)])
        out[f"{tag}_reward"] = np.array([vdd[v]["reward"] for v in range(N)], np.float64)
        out[f"{tag}_existence"] = np.array([vdd[v]["existence"] for v in range(N)], np.float64)
        out[f"{tag}_nearest"] = np.array([vdd[v]["nearest_dist"] for v in range(N)], np.float64)
        out[f"{tag}_gt_nearest"] = np.array([vdd[v]["gt_nearest_dist"] for v in range(N)], np.float64)
        out[f"{tag}_goal"] = np.array([[*goal_dict[v]["pos"], goal_dict[v]["heading"], goal_dict[v]["speed"]] for v in range(N)])
        out[f"{tag}_goal_achieved"] = np.array(ev.goal_achieved_all[n0[0]:])
        out[f"{tag}_coll_off"] = np.array([ev.collision_rate_scenario[n0[1]:], ev.offroad_rate_scenario[n0[1]:]]).reshape(2, -1)
        out[f"{tag}_ade_fde"] = np.array([ev.ades_all[n0[2]:], ev.fdes_all[n0[2]:]])
    m, _ = ev.compute_metrics()
    out["metric_names"] = np.array(list(m.keys()))
    out["metric_values"] = np.array([m[k] for k in m])
    for k in ("lin_speed_sim_all", "lin_speed_gt_all", "ang_speed_sim_all", "ang_speed_gt_all", "accel_sim_all", "accel_gt_all",
              "nearest_dist_sim_all", "nearest_dist_gt_all"):
        out["samples_" + k[:-4]] = np.concatenate(getattr(ev, k), 0)[:, 0]
    print({k: float(v) for k, v in m.items()})
    save("metrics", **out)


def interesting_cases():
    """(tag, vehicles, history_steps, steps, seed for the goals / logs, seed handed to `random`)."""
    return (("a", 14, 10, 90, 1, 0), ("b", 24, 10, 90, 2, 5), ("c", 9, 1, 90, 3, 2), ("none", 6, 10, 90, 4, 0))


def gen_interesting():
    """eval_mode one_agent / two_agent: the reference's find_interesting_agent / find_interesting_pair
    (policy_evaluator.py:308-414) on goals and logs with close goals, short logs, early exits and stationary vehicles."""
    import random as _random
    import types as _t
    _install_evaluator_stubs()
    from evaluators.policy_evaluator import PolicyEvaluator
    out = {}
    for tag, N, hist, steps, seed, rseed in interesting_cases():
        rng = np.random.default_rng(seed)
        goals = rng.uniform(-40, 40, (N, 2)) if tag != "none" else np.arange(2 * N, dtype=np.float64).reshape(N, 2) * 30.0
        for k in range(1, N if tag != "none" else 0, 3):           # clusters of close goals (some exactly equal: dist > 0 filter)
            goals[k] = goals[k - 1] + (rng.uniform(-6, 6, 2) if k % 2 else 0.0)
        traj = np.zeros((N, steps + 1, 6))
        traj[..., :2] = goals[:, None] + rng.normal(0, 20, (N, 1, 2)) * np.linspace(1, 0, steps + 1)[None, :, None]
        traj[..., 4] = 1.0
        for v in range(N):
            r = rng.random()
            if r < 0.35:
                traj[v, int(rng.integers(30, steps)):, 4] = 0.0     # leaves the log: goal relocates, goal timestep moves
            elif r < 0.45:
                traj[v, :int(rng.integers(hist + 5, 50)), 4] = 0.0  # appears late: fewer logged steps after the history
        moving = [v for v in range(N) if rng.random() < 0.8]
        ev = PolicyEvaluator.__new__(PolicyEvaluator)
        ev.cfg = spec.make_cfg()
        ev.steps, ev.history_steps, ev.vehicles_to_evaluate = steps, hist, list(moving)
        vehs = [_t.SimpleNamespace(getID=(lambda v=v: v), target_position=_XY(float(goals[v, 0]), float(goals[v, 1]))) for v in range(N)]
        gt = {v: {"traj": traj[v]} for v in range(N)}
        picks_a, picks_p = [], []
        for draw in range(6):
            _random.seed(rseed + draw)
            a = ev.find_interesting_agent(vehs, dict(gt))
            _random.seed(rseed + draw)
            pr = ev.find_interesting_pair(vehs, dict(gt))
            picks_a.append(-1 if a is None else a)
            picks_p.append([-1, -1] if pr is None else pr)
        out[f"{tag}_goals"], out[f"{tag}_traj"], out[f"{tag}_moving"] = goals, traj, np.array(moving)
        out[f"{tag}_cfg"] = np.array([hist, steps, rseed])
        out[f"{tag}_agent"], out[f"{tag}_pair"] = np.array(picks_a), np.array(picks_p)
        print(tag, "agents", picks_a, "pairs", picks_p)
    save("interesting", **out)


def gen_state_dict():
    """Names and shapes of the reference modules' state_dict() (the `state_dict` of the Lightning checkpoint that
    CtRLSim.load_from_checkpoint reads, models/ctrl_sim.py:19-25, eval_sim.py:52)."""
    cfg = spec.make_cfg()
    d = spec.Dims(cfg)
    sd = ref_shims.build_reference_model(cfg, weights.generate(d, 0)).state_dict()
    names = list(sd.keys())
    save("state_dict", ctrl_sim_names=np.array(names), ctrl_sim_ndim=np.array([sd[n].dim() for n in names]),
         ctrl_sim_shapes=np.array([list(sd[n].shape) + [0] * (4 - sd[n].dim()) for n in names]))


# --------------------------------------------------------------------------------------------- (f3) preprocessed dataset
def export_scene(scn, states, coll_unused, actions, rewards, existence, goals):
    """A simulated scene in the evaluators' export format (policy_evaluator.py:559-570: objects with per-step lists + roads from
    get_road_data) built from a recorded rollout."""
    inv = {v: k for k, v in scenarios.ROAD_TYPES.items()}
    N, T1 = states.shape[:2]
    objs = []
    for v in range(N):
        objs.append({"position": [{"x": float(states[v, t, 0]), "y": float(states[v, t, 1])} for t in range(T1)],
                     "velocity": [{"x": float(states[v, t, 2]), "y": float(states[v, t, 3])} for t in range(T1)],
                     "heading": [float(states[v, t, 4]) for t in range(T1)],
                     "existence": [float(e) for e in existence[v]],
                     "acceleration": [float(actions[v, t, 0]) if t < T1 - 1 else 0 for t in range(T1)],
                     "steering": [float(actions[v, t, 1]) if t < T1 - 1 else 0 for t in range(T1)],
                     "reward": [[float(x) for x in rewards[v, t]] for t in range(T1)],
                     "goal_position": {"x": float(goals[v, 0]), "y": float(goals[v, 1])},
                     "goal_heading": float(goals[v, 2]), "goal_speed": float(goals[v, 3]),
                     "width": float(scn.width[v]), "length": float(scn.length[v]), "type": "vehicle"})
    roads = []
    for pl, ty in zip(scn.road_points, scn.road_types):
        n = int(pl[:, 2].sum())
        roads.append({"geometry": [{"x": float(q[0]), "y": float(q[1])} for q in pl[:n]], "type": inv[int(np.argmax(ty))]})
    return {"name": "synthetic", "objects": objs, "roads": roads}


def gen_preprocessed():
    """The reference's dataset code on simulated scenes: RLWaymoDatasetCtRLSim.get_data in preprocessing mode writes the
    *_physics.pkl dictionary (extract_rawdata, distance rewards), and RLWaymoDataset.get in eval mode reads it back as
    {'rtgs', 'road_points', 'road_types'} — what Evaluator.load_preprocessed_data hands the evaluators and policies."""
    import json as _json
    import pickle
    import tempfile
    import types as _t
    _install_evaluator_stubs()
    for name in ("matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            sys.modules[name] = _t.ModuleType(name)
    from datasets.rl_waymo.dataset_ctrl_sim import RLWaymoDatasetCtRLSim
    g, gm = np.load(os.path.join(GOLD, "closed_loop.npz")), np.load(os.path.join(GOLD, "metrics.npz"))
    out = {}
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "raw", "test"))
    cfg = spec.make_cfg(**LOOP)
    d = spec.Dims(cfg)
    for tag in ("a", "b", "c"):
        rc = g[f"{tag}_recipe"]
        scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP,
                                      extent=float(rc[4]))
        ex = gm[f"{tag}_existence"]
        data = export_scene(scn, g[f"{tag}_states"], None, g[f"{tag}_actions"], gm[f"{tag}_reward"], ex, gm[f"{tag}_goal"])
        with open(os.path.join(tmp, "raw", "test", f"scene_{tag}.json"), "w") as fh:
            _json.dump(data, fh)
    w = cfg.dataset.waymo
    w.dataset_path, w.preprocess_dir = os.path.join(tmp, "raw"), os.path.join(tmp, "pre")
    w.preprocess, w.preprocess_real_data = False, True
    writer = RLWaymoDatasetCtRLSim(cfg, split_name="test", mode="eval")
    for i, f in enumerate(writer.files):
        with open(f) as fh:
            writer.get_data(_json.load(fh), i)
    w.preprocess, w.preprocess_real_data = True, False
    reader = RLWaymoDatasetCtRLSim(cfg, split_name="test", mode="eval")
    assert len(reader.files) == 3
    for i, f in enumerate(reader.files):
        tag = os.path.basename(f)[6]
        with open(f, "rb") as fh:
            pk = pickle.load(fh)
        dd = reader.get(i)
        for k in ("ag_data", "ag_actions", "ag_types", "last_exist_timesteps", "veh_edge_dist_rewards", "veh_veh_dist_rewards",
                  "ag_rewards", "ag_goals", "road_points", "road_types"):
            out[f"{tag}_pkl_{k}"] = np.asarray(pk[k])
        out[f"{tag}_pkl_filtered_ag_ids"] = np.asarray(pk["filtered_ag_ids"])
        out[f"{tag}_rtgs"] = dd["rtgs"]
        assert np.array_equal(dd["road_points"], pk["road_points"])
        print(tag, "rtgs", dd["rtgs"].shape, "rtg range", dd["rtgs"].min(), dd["rtgs"].max())
    save("preprocessed", **out)


def gen_ingest_gt():
    """The reference's PYTHON layer over a scenario file (utils/sim.py:20-79: get_ground_truth_states, get_road_data) run on a
    REPLAY of a Nocturne-format file: a stand-in `Simulation` whose expert-controlled vehicles walk through the file's own
    position / heading / velocity lists (what Scenario::LoadObjects + expert control do in C++, nocturne/cpp/src/scenario.cc —
    that C++ cannot be built here, so this pins the row layout, the existence rule (x != -10000), the steps + 1 length and the
    road-data ordering of the Python layer, not the C++ loader itself)."""
    import json as _json
    import tempfile
    import types as _t
    _install_evaluator_stubs()
    from ctrlsim_amd import ingest
    from ctrlsim_amd.scenarios import standin_log
    import utils.sim as ref_sim
    scn = scenarios.make_scenario(91, 0, n_agents=6, n_polylines=9, n_points=10, extent=30.0)
    steps = 20
    log = standin_log(scn, steps)
    log[2]["traj"][13:, 4] = 0.0                              # leaves the log
    log[4]["traj"][6:, 4] = 0.0
    data = ingest.scenario_to_nocturne_json(scn, log, name="replay")
    data["roads"].append({"geometry": [{"x": 3.25, "y": -7.5}], "type": "stop_sign"})
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "replay.json"), "w") as fh:
        _json.dump(data, fh)

    RT = scenarios.ROAD_TYPES

    class _Veh:
        def __init__(self, i, obj, sim):
            self.i, self.o, self.sim, self.expert_control = i, obj, sim, False
        def getID(self): return self.i
        def _t(self): return min(self.sim.t, len(self.o["position"]) - 1)
        def getPosition(self): p = self.o["position"][self._t()]; return _XY(np.float32(p["x"]), np.float32(p["y"]))
        def getHeading(self):                                   # NormalizeAngle(Radians(deg)) in float (scenario.cc:934-935)
            r = np.float32(np.float64(np.float32(self.o["heading"][self._t()])) / 180.0 * np.pi)
            r = np.float32(np.fmod(np.float64(r), 2.0 * np.pi))
            return np.float32(np.float64(r) - 2 * np.pi) if r > np.pi else (np.float32(np.float64(r) + 2 * np.pi) if r < -np.pi else r)
        def getSpeed(self): v = self.o["velocity"][self._t()]; return np.float32(np.sqrt(np.float32(v["x"]) ** 2 + np.float32(v["y"]) ** 2))
        def getGoalPosition(self): g = self.o["goalPosition"]; return _XY(np.float32(g["x"]), np.float32(g["y"]))
        def getType(self): return _t.SimpleNamespace(value=1)
        def getLength(self): return np.float32(self.o["length"])
        def getWidth(self): return np.float32(self.o["width"])

    class _Line:
        def __init__(self, road): self.road, self.road_type = road, RT[road["type"]]
        def geometry_points(self): return [_XY(np.float32(p["x"]), np.float32(p["y"])) for p in self.road["geometry"]]

    class _Stop:
        def __init__(self, road): self.road = road
        def position(self): p = self.road["geometry"][0]; return _XY(np.float32(p["x"]), np.float32(p["y"]))

    class _Sim:
        def __init__(self, scenario_path, config):
            with open(scenario_path) as fh:
                self.data = _json.load(fh)
            self.t = 0
            self.vehs = [_Veh(i, o, self) for i, o in enumerate(self.data["objects"]) if o["valid"][0]]
        def getScenario(self): return self
        def vehicles(self): return self.vehs
        def getObjectsThatMoved(self): return self.vehs
        def getRoadLines(self): return [_Line(r) for r in self.data["roads"] if r["type"] != "stop_sign"]
        def stop_signs(self): return [_Stop(r) for r in self.data["roads"] if r["type"] == "stop_sign"]
        def step(self, dt): self.t += 1
        def reset(self): self.t = 0

    ref_sim.Simulation = _Sim
    cfg = spec.make_cfg()
    gt = ref_sim.get_ground_truth_states(cfg, tmp, ["replay.json"], 0, 0.1, steps)
    road_data = ref_sim.get_road_data(_Sim(os.path.join(tmp, "replay.json"), None))
    ids = sorted(gt.keys())
    out = {"json": np.array(_json.dumps(data)), "ids": np.array(ids),
           "traj": np.array([gt[i]["traj"] for i in ids], np.float64), "type": np.array([gt[i]["type"] for i in ids], np.float64),
           "road_types": np.array([r["type"] for r in road_data]),
           "road_n": np.array([1 if isinstance(r["geometry"], dict) else len(r["geometry"]) for r in road_data]),
           "road_xy": np.array([[q["x"], q["y"]] for r in road_data
                                for q in ([r["geometry"]] if isinstance(r["geometry"], dict) else r["geometry"])], np.float64)}
    print("gt", out["traj"].shape, "roads", len(road_data))
    save("ingest_gt", **out)


# --------------------------------------------------------------------------------------------- G4 features
def gen_features():
    """Reference get_data() on hand-built policy buffers: exercises select_relevant_agents (first call and
    persisted-set path incl. an agent leaving the 60 m disc), discretize_*, normalize_scene (P_all < P padding
    and P_all > P nearest-polyline selection), grouping."""
    ref_shims.install()
    from policies.autoregressive_policy import AutoregressivePolicy
    out = {}
    for tag, over, n_ag, n_pl in (("small", LOOP, 10, 20), ("full", {}, 30, 260), ("wide", {}, 64, 512)):
        cfg = spec.make_cfg(**over)
        d = spec.Dims(cfg)
        dset = ref_shims.build_reference_dataset(cfg)

        class _M:  # policy only needs .cfg and .eval()
            def eval(self):
                return self
        m = _M()
        m.cfg = cfg
        pol = AutoregressivePolicy(cfg=cfg, model_path="", model=m, use_rtg=True, predict_rtgs=True,
                                   discretize_rtgs=True, real_time_rewards=False, privileged_return=False,
                                   max_return=False, min_return=False,
                                   key_dict={"next_acceleration": "next_acceleration",
                                             "next_steering": "next_steering", "rtgs": "rtgs"},
                                   tilt_dict={"tilt": True, "goal_tilt": 0, "veh_veh_tilt": 0, "veh_edge_tilt": 0},
                                   name="ctrl_sim", action_temperature=1.0, nucleus_sampling=False,
                                   nucleus_threshold=0.8)
        scn = scenarios.make_scenario(11, 0, n_agents=n_ag, n_polylines=n_pl, n_points=d.NP,
                                      extent=70.0 if tag != "small" else 45.0)
        bufs = synth_inputs.synth_policy_buffers(scn, cfg, seed=5)
        pol.reset({i: None for i in range(scn.N)})
        for k in ("states", "types", "actions", "rtgs", "goals", "timesteps"):
            getattr(pol, k)[:] = bufs[k]
        gt = {i: {"traj": np.ones((91, 6))} for i in range(scn.N)}
        preproc = {"road_points": scn.road_points.astype(np.float64), "road_types": scn.road_types.copy()}
        for t in (0, 3, d.T + 5):
            md, dead, idx_dicts, veh_ids = pol.get_data(gt, preproc, dset, list(range(scn.N)), t)
            foc = list(md.keys())
            out[f"{tag}_t{t}_focals"] = np.array(foc)
            for gi, f in enumerate(foc):
                a, mp = md[f]["agent"], md[f]["map"]
                pre = f"{tag}_t{t}_g{gi}_"
                out[pre + "ids"] = np.array(sorted(idx_dicts[f].keys()))
                out[pre + "members"] = np.array(veh_ids[f])
                keep_full = (tag == "small") or gi == 0
                if keep_full:
                    out[pre + "agent_states"] = a.agent_states.numpy()[0]
                    out[pre + "goals"] = a.goals.numpy()[0]
                    out[pre + "actions"] = a.actions.numpy()[0]
                    out[pre + "rtgs"] = a.rtgs.numpy()[0]
                    out[pre + "types"] = a.agent_types.numpy()[0]
                    out[pre + "timesteps"] = a.timesteps.numpy()[0]
                    rp = mp.road_points.numpy()[0]
                    out[pre + "road_points"] = rp if tag == "small" else rp[:, ::25].copy()
                    out[pre + "road_types"] = mp.road_types.numpy()[0]
                    out[pre + "road_points_sum"] = rp.sum(axis=(1, 2))
        out[f"{tag}_recipe"] = np.array([11, 0, n_ag, n_pl, 70.0 if tag != "small" else 45.0, 5])
    save("features", **out)


# --------------------------------------------------------------------------------------------- G5 sampling
def gen_sampling():
    """Reference sampling code path (process_predicted_rtg + predict's action block) on random logits."""
    ref_shims.install()
    from policies.policy import Policy
    import torch.nn.functional as F
    cfg = spec.make_cfg()
    d = spec.Dims(cfg)
    dset = ref_shims.build_reference_dataset(cfg)
    rs = np.random.RandomState(2)
    n = 64
    rtg_logits = (rs.normal(0, 2.0, (n, d.R * d.C))).astype(np.float32)
    act_logits = (rs.normal(0, 2.0, (n, d.V))).astype(np.float32)
    tilts = np.array([(0, 0, 0), (10, -10, 5), (-20, 30, 0)], np.float64)
    out = dict(rtg_logits=rtg_logits, act_logits=act_logits, tilts=tilts)
    for ti, tl in enumerate(tilts):
        bins = np.zeros((n, 3), np.int64)
        marg = np.zeros((n, 3))
        tilt_logits = torch.from_numpy(dset.get_tilt_logits(*tl))
        for i in range(n):
            lg = torch.from_numpy(rtg_logits[i]).reshape(d.R, d.C)
            for c in range(3):
                dis = F.softmax(lg[:, c] + tilt_logits[:, c], dim=0)
                q = weights.exp_noise(9, 0, 0, i, c, d.R)
                ratio = dis / torch.from_numpy(q).to(dis.dtype)
                bins[i, c] = int(torch.argmax(ratio))
                t2 = torch.topk(ratio, 2).values
                marg[i, c] = float((t2[0] - t2[1]) / t2[0])
        out[f"rtg_bins_tilt{ti}"] = bins
        out[f"rtg_margin_tilt{ti}"] = marg
    for tag, temp, nucleus in (("t1", 1.0, False), ("t15", 1.5, False), ("nuc", 1.0, True), ("nuc_t07", 0.7, True)):
        toks = np.zeros(n, np.int64)
        marg = np.zeros(n)
        for i in range(n):
            next_action_logits = torch.from_numpy(act_logits[i])
            if nucleus:  # same op sequence as autoregressive_policy.py:217-231
                action_probs = F.softmax(next_action_logits / temp, dim=0)
                sorted_probs, sorted_indices = torch.sort(action_probs, descending=True)
                cum_probs = torch.cumsum(sorted_probs, dim=-1)
                sel = cum_probs < 0.8
                sel = torch.cat([sel.new_ones(sel.shape[:-1] + (1,)), sel[..., :-1]], dim=-1)
                new_probs = sorted_probs[sel]
                new_probs /= new_probs.sum()
                dis = torch.zeros_like(next_action_logits)
                dis[sorted_indices[sel]] = new_probs
            else:
                dis = F.softmax(next_action_logits / temp, dim=0)
            q = weights.exp_noise(9, 0, 0, i, 3, d.V)
            ratio = dis / torch.from_numpy(q)
            toks[i] = int(torch.argmax(ratio))
            t2 = torch.topk(ratio, 2).values
            marg[i] = float((t2[0] - t2[1]) / t2[0])
        out[f"act_tok_{tag}"] = toks
        out[f"act_margin_{tag}"] = marg
    # agreement of the race formulation with the real torch.multinomial under a seeded CPU generator
    g = torch.Generator().manual_seed(1234)
    p = F.softmax(torch.from_numpy(act_logits[:32]), dim=1)
    agree = 0
    for i in range(32):
        g2 = torch.Generator().manual_seed(1000 + i)
        a = int(torch.multinomial(p[i], 1, generator=g2))
        g3 = torch.Generator().manual_seed(1000 + i)
        q = torch.empty_like(p[i]).exponential_(1, generator=g3)
        agree += int(a == int(torch.argmax(p[i] / q)))
    out["multinomial_agreement"] = np.array([agree, 32])
    print("multinomial == argmax(p/q):", agree, "/ 32")
    save("sampling", **out)


# --------------------------------------------------------------------------------------------- G6/G7/G9/G10
def gen_physics():
    rs = np.random.RandomState(4)
    n, steps = 8, 20
    L = rs.uniform(4, 5.5, n).astype(np.float32); W = rs.uniform(1.8, 2.3, n).astype(np.float32)
    x = (np.arange(n) * 40.0).astype(np.float32); y = rs.uniform(-5, 5, n).astype(np.float32)
    h = rs.uniform(-np.pi, np.pi, n).astype(np.float32); v = rs.uniform(0, 15, n).astype(np.float32)
    v[1] = 49.0
    acts = np.stack([rs.uniform(-10, 10, (steps, n)), rs.uniform(-0.7, 0.7, (steps, n))], -1)
    acts[:, 0] = (-10.0, 0.0)            # brake to zero, then Box2D auto-sleep after 0.5 s
    acts[:, 1] = (10.0, 5e-8)            # |steer| < 1e-7 branch, 50 m/s clamp (patched b2_maxTranslation)
    acts[:, 2, 0] = 0.0                  # Brake(0) is a no-op: previous throttle/brake persist
    acts[5:, 3] = (-0.0005, 0.7)         # Brake(|a|<1e-3) no-op after throttle
    segs = np.concatenate([np.stack([x - 10, y + 3, x + 30, y + 4], 1), np.stack([x + 5, y - 30, x + 6, y + 30], 1)])
    sim = RefSim(L, W, x, y, h, v, segs)
    traj = np.zeros((steps + 1, n, 6), np.float32); cv = np.zeros((steps + 1, n), np.uint8); ce = cv.copy()
    traj[0], cv[0], ce[0] = sim.state()
    for t in range(steps):
        for i in range(n):
            sim.set_action(i, acts[t, i, 0], acts[t, i, 1])
        sim.step(0.1)
        traj[t + 1], cv[t + 1], ce[t + 1] = sim.state()
    sim.close()
    out = dict(L=L, W=W, x=x, y=y, h=h, v=v, acts=acts, segs=segs.astype(np.float32), traj=traj, coll_veh=cv, coll_edge=ce)
    # contact case (tier-2, informational): two cars driving into each other
    sim = RefSim([4.5, 4.5], [2.0, 2.0], [0.0, 12.0], [0.0, 0.3], [0.0, np.pi], [8.0, 8.0])
    ct = np.zeros((steps + 1, 2, 6), np.float32); ccv = np.zeros((steps + 1, 2), np.uint8)
    ct[0], ccv[0], _ = sim.state()
    for t in range(steps):
        sim.set_action(0, 2.0, 0.0); sim.set_action(1, 2.0, 0.0)
        sim.step(0.1)
        ct[t + 1], ccv[t + 1], _ = sim.state()
    sim.close()
    out.update(contact_traj=ct, contact_coll_veh=ccv)
    save("physics", **out)


def contact_scene(kind, seed):
    """Scripted scenes in which vehicle boxes run into each other (used by gen_contacts and the tests)."""
    r = np.random.RandomState(seed)
    if kind == "headon":
        n = 2
        x = np.array([-10, 10], np.float32); y = np.array([0, r.uniform(-1.5, 1.5)], np.float32)
        h = np.array([r.uniform(-0.2, 0.2), np.pi + r.uniform(-0.2, 0.2)], np.float32)
        v = r.uniform(3, 12, n).astype(np.float32)
    elif kind == "tbone":
        n = 2
        x = np.array([-8, r.uniform(-1, 1)], np.float32); y = np.array([0, -9], np.float32)
        h = np.array([r.uniform(-0.1, 0.1), np.pi / 2 + r.uniform(-0.1, 0.1)], np.float32)
        v = r.uniform(5, 10, n).astype(np.float32)
    elif kind == "crowd":                   # cars converging on one point: islands of three and more bodies
        n = 8 + 4 * (seed % 3)
        ang = np.linspace(0, 2 * np.pi, n, endpoint=False) + r.uniform(-0.1, 0.1, n)
        rad = r.uniform(12, 22, n)
        x = (rad * np.cos(ang)).astype(np.float32); y = (rad * np.sin(ang)).astype(np.float32)
        h = (ang + np.pi + r.uniform(-0.15, 0.15, n)).astype(np.float32)
        v = r.uniform(4, 10, n).astype(np.float32)
    elif kind == "dense":                   # random placement in a small lot: boxes overlap from the start, large islands
        n = 32 + 16 * (seed % 2)
        x = r.uniform(-25, 25, n).astype(np.float32); y = r.uniform(-25, 25, n).astype(np.float32)
        h = r.uniform(-np.pi, np.pi, n).astype(np.float32)
        v = r.uniform(2, 10, n).astype(np.float32)
    else:                                   # "pairs": four separate two-car encounters in one world
        n = 8
        x = np.zeros(n, np.float32); y = np.zeros(n, np.float32); h = np.zeros(n, np.float32)
        for k in range(4):
            cx, cy = 60.0 * k, 40.0 * (k % 2)
            a = r.uniform(0, 2 * np.pi)
            x[2 * k], y[2 * k] = cx - 9 * np.cos(a), cy - 9 * np.sin(a)
            x[2 * k + 1], y[2 * k + 1] = cx + 9 * np.cos(a) + r.uniform(-1, 1), cy + 9 * np.sin(a) + r.uniform(-1, 1)
            h[2 * k], h[2 * k + 1] = a + r.uniform(-0.2, 0.2), a + np.pi + r.uniform(-0.2, 0.2)
        v = r.uniform(4, 11, n).astype(np.float32)
    L = r.uniform(4, 5.5, n).astype(np.float32); W = r.uniform(1.8, 2.3, n).astype(np.float32)
    steps = 40
    acts = np.stack([r.uniform(-3, 3, (steps, n)), r.uniform(-0.3, 0.3, (steps, n))], -1)
    return dict(L=L, W=W, x=x, y=y, h=h, v=v, acts=acts, segs=np.zeros((1, 4), np.float32) + 1e6)


def run_scripted(sim_cls, sc):
    sim = sim_cls(sc["L"], sc["W"], sc["x"], sc["y"], sc["h"], sc["v"], sc["segs"])
    steps, n = sc["acts"].shape[:2]
    traj = np.zeros((steps + 1, n, 6), np.float32); cv = np.zeros((steps + 1, n), np.uint8); body = np.zeros((steps + 1, n, 6), np.float32)
    traj[0], cv[0], _ = sim.state(); body[0] = sim.body()
    for t in range(steps):
        for i in range(n):
            sim.set_action(i, sc["acts"][t, i, 0], sc["acts"][t, i, 1])
        sim.step(0.1)
        traj[t + 1], cv[t + 1], _ = sim.state(); body[t + 1] = sim.body()
    sim.close()
    return traj, cv, body


def gen_contacts():
    """Vehicles colliding, through the REAL FreeCar + Box2D (contact solver active): trajectories + body velocities."""
    out = {}
    cases = [("headon", 3), ("headon", 10), ("tbone", 0), ("tbone", 5), ("pairs", 1), ("pairs", 2), ("crowd", 0), ("crowd", 1),
             ("crowd", 5), ("dense", 0), ("dense", 1)]
    for k, (kind, seed) in enumerate(cases):
        sc = contact_scene(kind, seed)
        traj, cv, body = run_scripted(RefSim, sc)
        print(kind, seed, "vehicle-collision flags", int(cv.sum()))
        assert cv.sum() > 0
        for key, val in sc.items():
            out[f"c{k}_{key}"] = val
        out[f"c{k}_traj"] = traj; out[f"c{k}_coll_veh"] = cv; out[f"c{k}_body"] = body
    out["n_cases"] = np.array(len(cases))
    save("contacts", **out)


def gen_collision():
    geo = ref_geo()
    rs = np.random.RandomState(6)
    n = 400
    boxes = np.zeros((n, 2, 4, 2), np.float32); segs = np.zeros((n, 4), np.float32)
    pp = np.zeros(n, np.uint8); ps = np.zeros(n, np.uint8)

    def box(cx, cy, L, W, th):
        c, s = np.float32(np.cos(np.float32(th))), np.float32(np.sin(np.float32(th)))
        hx = np.array([L / 2, -L / 2, -L / 2, L / 2], np.float32); hy = np.array([W / 2, W / 2, -W / 2, -W / 2], np.float32)
        return np.stack([hx * c - hy * s + np.float32(cx), hx * s + hy * c + np.float32(cy)], 1).astype(np.float32)
    for i in range(n):
        a = box(0, 0, rs.uniform(4, 5.5), rs.uniform(1.8, 2.3), rs.uniform(-np.pi, np.pi))
        b = box(rs.uniform(-6, 6), rs.uniform(-6, 6), rs.uniform(4, 5.5), rs.uniform(1.8, 2.3), rs.uniform(-np.pi, np.pi))
        sg = np.array([rs.uniform(-5, 5), rs.uniform(-5, 5), 0, 0], np.float32)
        sg[2:] = sg[:2] + rs.uniform(-6, 6, 2)
        if i % 50 == 0:
            sg[2:] = sg[:2]                       # degenerate segment -> Contains()
        boxes[i, 0], boxes[i, 1], segs[i] = a, b, sg
        pp[i] = geo.refgeo_poly_poly(np.ascontiguousarray(a), 4, np.ascontiguousarray(b), 4)
        ps[i] = geo.refgeo_poly_seg(np.ascontiguousarray(a), 4, sg)
    print("collision fixture: poly-poly hits", pp.sum(), "poly-seg hits", ps.sum())
    save("collision", boxes=boxes, segs=segs, poly_poly=pp, poly_seg=ps)


def gen_bicycle():
    sys.path.insert(0, ref_shims.REF)
    import importlib.util
    sp = importlib.util.spec_from_file_location("ref_bicycle_model", ref_shims.REF + "/nocturne/bicycle_model.py")
    mod = importlib.util.module_from_spec(sp)
    import matplotlib
    matplotlib.use("Agg")
    sp.loader.exec_module(mod)
    rs = np.random.RandomState(8)
    n = 1000
    nxt = np.stack([rs.uniform(-100, 100, n), rs.uniform(-100, 100, n), rs.uniform(-np.pi, np.pi, n),
                    rs.uniform(0, 20, n), rs.uniform(4, 5.5, n)], 1)
    prev = np.stack([nxt[:, 0] + rs.normal(0, 1, n), nxt[:, 1] + rs.normal(0, 1, n),
                     nxt[:, 2] + rs.normal(0, 0.2, n), np.abs(nxt[:, 3] + rs.normal(0, 1, n))], 1)
    prev[::10, 3] = -nxt[::10, 3]   # v_next + v_prev == 0 edge
    res = np.zeros((n, 2))
    for i in range(n):
        bm = mod.BicycleModel(x=nxt[i, 0], y=nxt[i, 1], theta=nxt[i, 2], vel=nxt[i, 3], L=nxt[i, 4], dt=0.1)
        a, s, _, _ = bm.backward(prev_pos=prev[i, :2], prev_theta=prev[i, 2], prev_vel=prev[i, 3])
        res[i] = (a, s)
    save("bicycle_backward", nxt=nxt, prev=prev, accel_steer=res)


# --------------------------------------------------------------------------------------------- planner vs adversary
PLANNER_TILT = (10.0, 10.0, 10.0)       # cfgs/policy/ctrl_sim_planner.yaml:5-7
ADVERSARY_TILT = (0.0, -10.0, 0.0)      # cfgs/policy/ctrl_sim_adversary.yaml:6-8


def pick_ego_adversary(scn):
    """Synthetic stand-in of the CAT dictionary (planner_adversary_evaluator.py:431-456): ego = vehicle 0, adversary = the
    vehicle nearest to it at t = 0."""
    d = np.hypot(scn.x - scn.x[0], scn.y - scn.y[0])
    d[0] = np.inf
    return 0, int(np.argmin(d))


def ref_planner_adversary(cfg, w, scn, steps, seed, history_steps):
    """evaluate_planner_adversary's inner loop (planner_adversary_evaluator.py:497-546) around two UNMODIFIED reference
    policies: the planner drives the ego, the adversary drives one other vehicle, everybody else (and both of them before
    history_steps - 1) replays the log through the reference's inverse bicycle model (evaluators/evaluator.py:160-193)."""
    ref_shims.install()
    from policies.autoregressive_policy import AutoregressivePolicy
    import importlib.util
    sp = importlib.util.spec_from_file_location("ref_bicycle_model", ref_shims.REF + "/nocturne/bicycle_model.py")
    bmod = importlib.util.module_from_spec(sp)
    import matplotlib
    matplotlib.use("Agg")
    sp.loader.exec_module(bmod)

    model = ref_shims.build_reference_model(cfg, w)
    dset = ref_shims.build_reference_dataset(cfg)

    def make(role, tilt):
        key_dict = {"next_acceleration": f"next_{role}_acceleration", "next_steering": f"next_{role}_steering",
                    "rtgs": f"{role}_rtgs"}
        tilt_dict = {"tilt": True, "goal_tilt": tilt[0], "veh_veh_tilt": tilt[1], "veh_edge_tilt": tilt[2]}
        return AutoregressivePolicy(cfg=cfg, model_path="", model=model, use_rtg=True, predict_rtgs=True,
                                    discretize_rtgs=True, real_time_rewards=False, privileged_return=False,
                                    max_return=False, min_return=False, key_dict=key_dict, tilt_dict=tilt_dict,
                                    name="ctrl_sim", action_temperature=1.0, nucleus_sampling=False, nucleus_threshold=0.8)
    planner, adversary = make("planner", PLANNER_TILT), make("adversary", ADVERSARY_TILT)
    N = scn.N
    ego, adv = pick_ego_adversary(scn)
    sim = RefSim(scn.length, scn.width, scn.x, scn.y, scn.heading, scn.speed, scn.edge_segments)
    vehs = [_FakeVeh(sim, i) for i in range(N)]
    gt = scenarios.standin_log(scn, steps)
    vdd = {}
    for i in range(N):
        vdd[i] = {"position": [], "velocity": [], "heading": [], "existence": [], "acceleration": [], "steering": [],
                  "timestep": [], "planner_rtgs": [], "next_planner_acceleration": 0., "next_planner_steering": 0.,
                  "adversary_rtgs": [], "next_adversary_acceleration": 0., "next_adversary_steering": 0.,
                  "goal_position": {"x": scn.goal_pos[i, 0], "y": scn.goal_pos[i, 1]},
                  "goal_heading": scn.goal_heading[i], "goal_speed": scn.goal_speed[i],
                  "width": scn.width[i], "length": scn.length[i], "type": "vehicle"}
    preproc = {"road_points": scn.road_points.astype(np.float64), "road_types": scn.road_types.copy()}
    patches = {"planner": _NoisePatch(seed, scn.index), "adversary": _NoisePatch(seed, scn.index)}
    orig = torch.multinomial
    states = np.zeros((N, steps + 1, 8))
    coll = np.zeros((N, steps + 1, 2), np.uint8)
    applied = np.zeros((N, steps, 2))
    rtg_cont = np.zeros((2, N, steps, 3))
    cur = {}

    def update(t):
        st, cv, ce = sim.state()
        cur["st"] = st
        for i in range(N):
            vdd[i]["position"].append({"x": st[i, 0], "y": st[i, 1]})
            vdd[i]["velocity"].append({"x": st[i, 4], "y": st[i, 5]})
            vdd[i]["heading"].append(st[i, 2])
            vdd[i]["timestep"].append(t)
            vdd[i]["existence"].append(1.0)
            states[i, t] = [st[i, 0], st[i, 1], st[i, 4], st[i, 5], st[i, 2], scn.length[i], scn.width[i], 1.0]
        coll[:, t, 0], coll[:, t, 1] = cv, ce

    def apply_gt_action(i, t):
        tr = gt[i]["traj"]
        st = cur["st"]
        bm = bmod.BicycleModel(x=tr[t + 1][0], y=tr[t + 1][1], theta=tr[t + 1][2], vel=tr[t + 1][3], L=tr[t + 1][-1], dt=0.1)
        a, s, _, _ = bm.backward(prev_pos=np.array([st[i, 0], st[i, 1]]), prev_theta=st[i, 2], prev_vel=st[i, 3])
        v = vehs[i]
        if a > 0.0:
            v.acceleration = a
        else:
            v.brake(np.abs(a))
        v.steering = s
        return [a, s]

    try:
        planner.reset(vdd)
        adversary.reset(vdd)
        for t in range(steps):
            update(t)
            planner.update_state(vdd, [ego], t)
            adversary.update_state(vdd, [adv], t)
            for role, pol, who in (("planner", planner, ego), ("adversary", adversary, adv)):
                patches[role].t = t
                torch.multinomial = patches[role]
                vdd = pol.predict(vdd, gt, preproc, dset, [who], t)
                torch.multinomial = orig
            for i in range(N):
                if t >= history_steps - 1 and i == ego:
                    _, act = planner.act(vehs[i], t, vdd)
                elif t >= history_steps - 1 and i == adv:
                    _, act = adversary.act(vehs[i], t, vdd)
                else:
                    act = apply_gt_action(i, t)
                vdd[i]["acceleration"].append(act[0])
                vdd[i]["steering"].append(act[1])
                applied[i, t] = act
                rtg_cont[0, i, t] = vdd[i]["planner_rtgs"][-1]
                rtg_cont[1, i, t] = vdd[i]["adversary_rtgs"][-1]
            sim.step(0.1)
        update(steps)
    finally:
        torch.multinomial = orig
        sim.close()
    tokens = dset.discretize_actions(applied.copy())
    return dict(states=states, coll=coll, actions=applied, tokens=tokens, rtg_cont=rtg_cont, ego_adv=np.array([ego, adv]),
                margins=np.array(patches["planner"].log + patches["adversary"].log))


def gen_planner_adversary():
    cfg = spec.make_cfg(**LOOP)
    d = spec.Dims(cfg)
    w = weights.generate(d, 0)
    out = {}
    for tag, idx0, n_ag, n_pl, extent, hist in (("a", 0, 8, 12, 25.0, 3), ("b", 1, 10, 9, 30.0, 1)):
        for idx in range(idx0, 50):     # first scene whose sampling races are all clear of ties ("b": and with a collision)
            scn = scenarios.make_scenario(9, idx, n_agents=n_ag, n_polylines=n_pl, n_points=d.NP, extent=extent)
            r = ref_planner_adversary(cfg, w, scn, 20, seed=3, history_steps=hist)
            if r["margins"].min() > 2e-4 and (tag == "a" or r["coll"][..., 0].sum() > 0):
                break
        print(tag, "ego/adv", r["ego_adv"], "veh-veh flags", r["coll"][..., 0].sum(), "min race margin", r["margins"].min())
        for k, v in r.items():
            out[f"{tag}_{k}"] = v
        out[f"{tag}_recipe"] = np.array([9, idx, n_ag, n_pl, extent, 3, hist])
    save("planner_adversary", **out)


# --------------------------------------------------------------------------------------------- ingest (road chunking)
def ingest_road_data(seed=12):
    """A road list in the get_road_data layout that exercises every branch of RLWaymoDataset.get_roads."""
    rs = np.random.RandomState(seed)

    def line(n, kind):
        xy = np.cumsum(rs.normal(0, 1, (n, 2)), 0) + rs.uniform(-50, 50, 2)
        return {"geometry": [{"x": float(np.float32(a)), "y": float(np.float32(b))} for a, b in xy], "type": kind}
    return [line(250, "lane"), line(100, "road_edge"), line(3, "crosswalk"),
            {"geometry": {"x": 4.5, "y": -7.25}, "type": "stop_sign"}, line(1, "speed_bump"), line(101, "road_line"),
            line(37, "road_edge"), line(200, "lane")]


def gen_ingest():
    import json
    cfg = spec.make_cfg()
    dset = ref_shims.build_reference_dataset(cfg)
    road_data = ingest_road_data()
    pts, types, edges = dset.get_roads({"roads": road_data})
    print("chunks", pts.shape, "edge polylines", [e.shape for e in edges])
    save("ingest", road_json=np.frombuffer(json.dumps(road_data).encode(), np.uint8), road_points=pts, road_types=types,
         n_edges=np.array(len(edges)), **{f"edge{i}": e for i, e in enumerate(edges)})


# --------------------------------------------------------------------------------------------- IL / Trajeglish variants
def variant_weights(cfg, w):
    """The reference IL / Trajeglish modules have no predict_rtg / predict_future_states heads (cfgs/model/{il,trajeglish}.yaml); with
    use_map = False the Encoder has no MapEncoder (modules/encoder.py:18)."""
    if not cfg.model.get("use_map", True):
        w = {k: v for k, v in w.items() if "map_encoder." not in k}
    if not (cfg.model.get("il", False) or cfg.model.get("trajeglish", False)):
        return w
    return {k: v for k, v in w.items() if not k.startswith(("decoder.predict_rtg", "decoder.predict_future_states"))}


def variant_cfg(name, **over):
    return spec.make_cfg(**{f"model__{name}": True, "model__predict_rtg": False, "model__predict_future_states": False}, **over)


def gen_variants():
    """cfgs/model/il.yaml and trajeglish.yaml through the UNMODIFIED reference modules / policy: masks, logits, closed loop."""
    ref_shims.install()
    from utils.train_utils import get_causal_mask
    out = {}
    for name, K in (("il", 2), ("trajeglish", 1), ("decision_transformer", 3)):
        # masks (small, full array) and the closed form at the real size
        cfg_t = variant_cfg(name, **TINY)
        out[f"{name}_mask_tiny"] = (get_causal_mask(cfg_t, 4, K) == 0).numpy()
        sidx = 1 if name == "decision_transformer" else 0
        # logits: tiny config in full, loop config at the last filled step
        for tag, over in (("tiny", TINY), ("loop", LOOP)):
            cfg = variant_cfg(name, **over)
            d = spec.Dims(cfg)
            w = weights.generate(d, 0)
            ref = ref_shims.build_reference_model(cfg, variant_weights(cfg, w))
            cm = model_oracle.causal_mask_closed_form(d.A, d.T, K, sidx)
            assert bool(((ref.decoder.causal_mask == 0) == cm).all()), "closed-form mask != get_causal_mask"
            for seed, t_fill in ((1, d.T), (2, max(1, d.T // 2))):
                inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=d.A - 1, n_polys=d.P - 1)
                if name == "decision_transformer":                    # continuous, normalised RTGs (autoregressive_policy.py:73-78)
                    inp["rtgs"] = synth_inputs.dt_rtgs(inp["rtgs"], seed)
                r = ref(synth_inputs.to_motion_data(inp), eval=True)
                assert set(r.keys()) == {"action_preds"}
                ap = r["action_preds"].detach().numpy()
                out[f"{name}_{tag}_s{seed}_action"] = ap if tag == "tiny" else ap[0, :, t_fill - 1]
                out[f"{name}_{tag}_s{seed}_recipe"] = np.array([seed, t_fill, d.A - 1, d.P - 1])
        if name == "decision_transformer":
            continue                                                  # its policy needs real-time rewards: see gen_dt_loop
        # closed loop, 14 steps (window T = 8 slides from step 8 on)
        cfg = variant_cfg(name, **LOOP)
        d = spec.Dims(cfg)
        w = weights.generate(d, 0)
        for idx in range(60):
            scn = scenarios.make_scenario(17, idx, n_agents=9, n_polylines=15, n_points=d.NP, extent=40.0)
            r = ref_closed_loop(cfg, w, scn, 14, seed=6, rtgs=False)
            if r["margins"].min() > 2e-4:
                break
        print(name, "scene", idx, "groups/step", r["n_groups"], "min race margin", r["margins"].min(), "veh-veh flags",
              r["coll"][..., 0].sum())
        for k in ("tokens", "states", "coll", "actions", "n_groups", "margins"):
            out[f"{name}_loop_{k}"] = r[k]
        out[f"{name}_loop_recipe"] = np.array([17, idx, 9, 15, 40.0, 6])
    save("variants", **out)


def gen_own_return():
    """cfg.model.attend_own_return_action = True (cfgs/model/base.yaml:15; utils/train_utils.py:114-129) through the UNMODIFIED reference:
    the mask itself, logits of the reference Encoder / Decoder built with that cfg (tiny dims in full, loop dims at the last filled step, both
    heads the policy reads) and a closed loop of the unmodified reference policy + real FreeCar / Box2D, 14 steps at the loop dims (the window
    T = 8 slides from step 8 on), tilts on."""
    ref_shims.install()
    from utils.train_utils import get_causal_mask
    out = {}
    own = {"model__attend_own_return_action": True}
    out["mask_tiny"] = (get_causal_mask(spec.make_cfg(**TINY, **own), 4, 3) == 0).numpy()
    for tag, over in (("tiny", TINY), ("loop", LOOP)):
        cfg = spec.make_cfg(**over, **own)
        d = spec.Dims(cfg)
        assert d.MASK_OWN
        w = weights.generate(d, 0)
        ref = ref_shims.build_reference_model(cfg, w)
        cm = model_oracle.causal_mask_closed_form(d.A, d.T, 3, 0, True)
        assert bool(((ref.decoder.causal_mask == 0) == cm).all()), "closed-form mask != get_causal_mask"
        for seed, t_fill in ((1, d.T), (2, max(1, d.T // 2))):
            inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=d.A - 1, n_polys=d.P - 1)
            r = ref(synth_inputs.to_motion_data(inp), eval=True)
            for head in ("action_preds", "rtg_preds"):
                v = r[head].detach().numpy()
                out[f"{tag}_s{seed}_{head}"] = v if tag == "tiny" else v[0, :, t_fill - 1]
            out[f"{tag}_s{seed}_recipe"] = np.array([seed, t_fill, d.A - 1, d.P - 1])
    cfg = spec.make_cfg(**LOOP, **own)
    d = spec.Dims(cfg)
    w = weights.generate(d, 0)
    for idx in range(60):
        scn = scenarios.make_scenario(19, idx, n_agents=9, n_polylines=15, n_points=d.NP, extent=40.0)
        r = ref_closed_loop(cfg, w, scn, 14, seed=5, tilt=(5.0, -10.0, 10.0))
        if r["margins"].min() > 2e-4:
            break
    print("own_return scene", idx, "groups/step", r["n_groups"], "min race margin", r["margins"].min(), "veh-veh flags", r["coll"][..., 0].sum())
    for k in ("tokens", "rtg_cont", "states", "coll", "actions", "n_groups", "margins"):
        out[f"loop_{k}"] = r[k]
    out["loop_recipe"] = np.array([19, idx, 9, 15, 40.0, 5, 5.0, -10.0, 10.0])
    # the same scene under the DEFAULT mask must differ (the fixture is not vacuous)
    r0 = ref_closed_loop(spec.make_cfg(**LOOP), w, scn, 14, seed=5, tilt=(5.0, -10.0, 10.0))
    out["loop_tokens_default_mask"] = r0["tokens"]
    print("tokens that differ from the default mask's rollout:", int((r0["tokens"] != r["tokens"]).sum()), "of", r["tokens"].size)
    save("own_return", **out)


MODEL_FLAG_CASES = {"no_actions": {"model__no_actions": True}, "no_map": {"model__use_map": False},
                    "no_init": {"model__encode_initial_state": False},
                    "no_actions_no_map": {"model__no_actions": True, "model__use_map": False},
                    "own_return_no_init": {"model__attend_own_return_action": True, "model__encode_initial_state": False}}


def gen_model_flags():
    """cfg.model.no_actions = True, use_map = False, encode_initial_state = False (cfgs/model/base.yaml:4,10; ctrl_sim.yaml:9;
    modules/encoder.py:18,84,129-130,155-170) through the UNMODIFIED reference: logits of the reference Encoder / Decoder built with each cfg
    (tiny dims: every token of both heads; loop dims: the slice the policy reads at the last filled step) and, for the three single switches,
    a closed loop of the unmodified reference policy + real FreeCar / Box2D (14 steps at the loop dims, through the window slide)."""
    ref_shims.install()
    out = {}
    for name, over in MODEL_FLAG_CASES.items():
        for tag, dims_over in (("tiny", TINY), ("loop", LOOP)):
            cfg = spec.make_cfg(**dims_over, **over)
            d = spec.Dims(cfg)
            assert d.FLAGS
            w = weights.generate(d, 0)
            ref = ref_shims.build_reference_model(cfg, variant_weights(cfg, w))
            for seed, t_fill in ((1, d.T), (2, max(1, d.T // 2))):
                inp = synth_inputs.random_context(d, seed, B=1, t_fill=t_fill, n_agents=d.A - 1, n_polys=d.P - 1)
                r = ref(synth_inputs.to_motion_data(inp), eval=True)
                for head in ("action_preds", "rtg_preds"):
                    v = r[head].detach().numpy()
                    out[f"{name}_{tag}_s{seed}_{head}"] = v if tag == "tiny" else v[0, :, t_fill - 1]
                out[f"{name}_{tag}_s{seed}_recipe"] = np.array([seed, t_fill, d.A - 1, d.P - 1])
        if name in ("no_actions_no_map", "own_return_no_init"):      # combinations: logits only
            continue
        cfg = spec.make_cfg(**LOOP, **over)
        d = spec.Dims(cfg)
        w = weights.generate(d, 0)
        for idx in range(60):
            scn = scenarios.make_scenario(23, idx, n_agents=9, n_polylines=15, n_points=d.NP, extent=40.0)
            r = ref_closed_loop(cfg, w, scn, 14, seed=6, tilt=(0.0, 5.0, -5.0))
            if r["margins"].min() > 2e-4:
                break
        r0 = ref_closed_loop(spec.make_cfg(**LOOP), w, scn, 14, seed=6, tilt=(0.0, 5.0, -5.0))
        print(name, "scene", idx, "groups/step", r["n_groups"], "min race margin", r["margins"].min(), "veh-veh flags", r["coll"][..., 0].sum(),
              "tokens that differ from the shipped cfg's rollout:", int((r0["tokens"] != r["tokens"]).sum()), "of", r["tokens"].size)
        for k in ("tokens", "rtg_cont", "states", "coll", "actions", "n_groups", "margins"):
            out[f"{name}_loop_{k}"] = r[k]
        out[f"{name}_loop_recipe"] = np.array([23, idx, 9, 15, 40.0, 6, 0.0, 5.0, -5.0])
        out[f"{name}_loop_tokens_shipped_cfg"] = r0["tokens"]
    save("model_flags", **out)


# --------------------------------------------------------------------------------------------- real-time (dense) rewards
def ref_dense_reward(dset, w, xy, exist, rewards, polys):
    """evaluators/evaluator.py:106-140, line by line, on the reference's dataset methods.  rewards [N, steps so far, 8].
    -> (dense_reward rows [N,3] as appended there, nearest_dist metric values [N], signed-edge reward [N,1])."""
    processed = rewards * exist[:, None, None]
    ag = xy[:, None, :].copy()
    edge = dset.compute_dist_to_nearest_road_edge_rewards(ag, polys) * exist[:, None]
    ag = np.concatenate([xy, exist[:, None]], 1)[:, None, :].copy()
    veh_raw = dset.compute_dist_to_nearest_vehicle_rewards(ag, normalize=False) * exist[:, None]
    nearest = veh_raw[:, 0] * w.max_veh_veh_distance
    veh = np.clip(veh_raw, 0.0, w.max_veh_veh_distance) / w.max_veh_veh_distance
    allr = dset.compute_rewards(ag, processed, edge, veh)
    allr = np.concatenate([allr[:, :, :1], allr[:, :, 3:]], -1)
    return allr[:, 0], nearest, edge


def gen_dense_reward():
    """The reference's own reward functions, called exactly as Evaluator.compute_dense_reward calls them
    (evaluators/evaluator.py:106-140), on random scenes: positions, existence, reward rows of several steps, road-edge polylines."""
    cfg = spec.make_cfg()
    dset = ref_shims.build_reference_dataset(cfg)
    w = cfg.dataset.waymo
    out = {}
    for case in range(4):
        rs = np.random.RandomState(40 + case)
        N, T = (1, 2) if case == 3 else (9, 3 + case)
        xy = rs.uniform(-30, 30, (N, 2))
        exist = (rs.uniform(size=N) > 0.25).astype(float)
        if case == 2:
            exist[:] = 0
            exist[0] = 1                                              # a single existing vehicle: nearest distance undefined
        rewards = rs.uniform(0, 1, (N, T, 8))
        rewards[..., [0, 1, 2, 6, 7]] = (rewards[..., [0, 1, 2, 6, 7]] > 0.6).astype(float)
        polys = []
        for k in range(3):
            n = [12, 2, 30][k]
            ang = np.sort(rs.uniform(0, 2 * np.pi, n))
            rad = rs.uniform(15, 25, n)
            p = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1) + rs.uniform(-10, 10, 2)
            if k == 2:
                p = np.concatenate([p, p[:1] + 0.1])                  # closed ring (cyclic branch)
            polys.append(p)
        allr, nearest, edge = ref_dense_reward(dset, w, xy, exist, rewards, polys)
        out[f"c{case}_xy"], out[f"c{case}_exist"], out[f"c{case}_rewards"] = xy, exist, rewards
        out[f"c{case}_npoly"] = np.array(len(polys))
        for k, p in enumerate(polys):
            out[f"c{case}_poly{k}"] = p
        out[f"c{case}_dense"] = allr
        out[f"c{case}_nearest_metric"] = nearest
        out[f"c{case}_edge_signed"] = edge[:, 0]
    save("dense_reward", **out)


def ref_closed_loop_dt(cfg, w, scn, steps, seed):
    """The Decision-Transformer policy of cfgs/policy/dt.yaml (real_time_rewards, max_return, continuous RTGs) in the loop of
    policy_evaluator.py:514-557 with its RTG bookkeeping (:122-153): the UNMODIFIED reference policy and model, the reference's
    own reward functions, the real FreeCar/Box2D; the per-vehicle compute_reward rows (utils/sim.py:83-141) are restated (only
    their goal / collision flags enter)."""
    ref_shims.install()
    from policies.autoregressive_policy import AutoregressivePolicy
    model = ref_shims.build_reference_model(cfg, w)
    dset = ref_shims.build_reference_dataset(cfg)
    wcfg = cfg.dataset.waymo
    pol = AutoregressivePolicy(cfg=cfg, model_path="", model=model, use_rtg=True, predict_rtgs=False, discretize_rtgs=False,
                               real_time_rewards=True, privileged_return=False, max_return=True, min_return=False,
                               key_dict={"next_acceleration": "next_acceleration", "next_steering": "next_steering", "rtgs": "rtgs"},
                               tilt_dict={"tilt": False, "goal_tilt": None, "veh_veh_tilt": None, "veh_edge_tilt": None},
                               name="dt", action_temperature=1.0, nucleus_sampling=False, nucleus_threshold=0.8)
    N = scn.N
    sim = RefSim(scn.length, scn.width, scn.x, scn.y, scn.heading, scn.speed, scn.edge_segments)
    vehs = [_FakeVeh(sim, i) for i in range(N)]
    polys = [np.asarray(pl[:int(pl[:, 2].sum()), :2], np.float64) for pl, ty in zip(scn.road_points, scn.road_types)
             if int(np.argmax(ty)) == 3]
    vdd = {i: {"position": [], "velocity": [], "heading": [], "existence": [], "acceleration": [], "steering": [], "timestep": [],
               "rtgs": [], "reward": [], "dense_reward": [], "next_acceleration": 0., "next_steering": 0.,
               "goal_position": {"x": scn.goal_pos[i, 0], "y": scn.goal_pos[i, 1]}, "goal_heading": scn.goal_heading[i],
               "goal_speed": scn.goal_speed[i], "width": scn.width[i], "length": scn.length[i], "type": "vehicle"} for i in range(N)}
    gt = {i: {"traj": np.ones((91, 6))} for i in range(N)}
    preproc = {"road_points": scn.road_points.astype(np.float64), "road_types": scn.road_types.copy()}
    to_eval = list(range(N))
    patch = _NoisePatch(seed, scn.index)
    orig = torch.multinomial
    torch.multinomial = patch
    states = np.zeros((N, steps + 1, 8)); coll = np.zeros((N, steps + 1, 2), np.uint8)
    applied = np.zeros((N, steps, 2)); rtg_raw = np.zeros((N, steps, 3)); dense_log = np.zeros((N, steps, 3))

    def update(t):
        st, cv, ce = sim.state()
        for i in range(N):
            d = vdd[i]
            d["position"].append({"x": st[i, 0], "y": st[i, 1]}); d["velocity"].append({"x": st[i, 4], "y": st[i, 5]})
            d["heading"].append(st[i, 2]); d["timestep"].append(t); d["existence"].append(1.0)
            states[i, t] = [st[i, 0], st[i, 1], st[i, 4], st[i, 5], st[i, 2], scn.length[i], scn.width[i], 1.0]
            if t == 0:                                                   # policy_evaluator.py:123-144 with max_return
                d["rtgs"].append(np.array([10.0, 90.0, 90.0]))
            else:
                d["rtgs"].append(d["rtgs"][-1] - d["dense_reward"][-1])
            reached = float(True) if (d["reward"] and d["reward"][-1][0]) else \
                float(np.linalg.norm(scn.goal_pos[i].astype(np.float64) - st[i, :2]) < 1.0)
            d["reward"].append([reached, 0.0, 0.0, 0.0, 0.0, 0.0, float(cv[i]), float(ce[i])])
        coll[:, t, 0], coll[:, t, 1] = cv, ce
        xy = np.array([[vdd[i]["position"][t]["x"], vdd[i]["position"][t]["y"]] for i in range(N)], np.float64)
        rew = np.array([vdd[i]["reward"] for i in range(N)], np.float64)
        dense, _, _ = ref_dense_reward(dset, wcfg, xy, np.ones(N), rew, polys)
        for i in range(N):
            vdd[i]["dense_reward"].append(dense[i])

    try:
        pol.reset(vdd)
        for t in range(steps):
            patch.t = t
            update(t)
            pol.update_state(vdd, to_eval, t)
            vdd = pol.predict(vdd, gt, preproc, dset, to_eval, t)
            for i in range(N):
                _, act = pol.act(vehs[i], t, vdd)
                vdd[i]["acceleration"].append(act[0]); vdd[i]["steering"].append(act[1])
                applied[i, t] = act
                rtg_raw[i, t] = vdd[i]["rtgs"][t]; dense_log[i, t] = vdd[i]["dense_reward"][t]
            sim.step(0.1)
        update(steps)
    finally:
        torch.multinomial = orig
        sim.close()
    return dict(tokens=dset.discretize_actions(applied.copy()), states=states, coll=coll, actions=applied, rtgs=rtg_raw,
                dense=dense_log, margins=np.array(patch.log))


def gen_dt_loop():
    cfg = variant_cfg("decision_transformer", **LOOP)
    d = spec.Dims(cfg)
    w = weights.generate(d, 0)
    for idx in range(60):
        scn = scenarios.make_scenario(19, idx, n_agents=9, n_polylines=15, n_points=d.NP, extent=40.0)
        r = ref_closed_loop_dt(cfg, w, scn, 14, seed=8)
        if r["margins"].min() > 2e-4:
            break
    print("dt scene", idx, "min race margin", r["margins"].min(), "veh-veh flags", r["coll"][..., 0].sum(), "rtg[0] range",
          r["rtgs"][..., 0].min(), r["rtgs"][..., 0].max(), "rtg[1] range", r["rtgs"][..., 1].min(), r["rtgs"][..., 1].max())
    out = {f"loop_{k}": v for k, v in r.items()}
    out["loop_recipe"] = np.array([19, idx, 9, 15, 40.0, 8])
    save("dt_loop", **out)


ALL = dict(model=gen_model, model_trained=gen_model_trained, closed_loop_trained=gen_closed_loop_trained, closed_loop_wide_trained=gen_closed_loop_wide_trained, features=gen_features, sampling=gen_sampling, physics=gen_physics,
           collision=gen_collision, closed_loop=gen_closed_loop, closed_loop_full=gen_closed_loop_full, closed_loop_wide=gen_closed_loop_wide, metrics=gen_metrics, interesting=gen_interesting, preprocessed=gen_preprocessed, ingest_gt=gen_ingest_gt, state_dict=gen_state_dict, bicycle=gen_bicycle, contacts=gen_contacts,
           planner_adversary=gen_planner_adversary, ingest=gen_ingest,
           variants=gen_variants, dense_reward=gen_dense_reward, dt_loop=gen_dt_loop, own_return=gen_own_return, model_flags=gen_model_flags)

if __name__ == "__main__":
    assert ref_shims.available(), "the reference tree is required to (re)generate golden vectors"
    torch.set_num_threads(8)
    names = sys.argv[1:] or list(ALL)
    for nme in names:
        print("==", nme)
        ALL[nme]()
