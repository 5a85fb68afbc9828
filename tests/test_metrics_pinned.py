"""E2 pinned: ctrlsim_amd.metrics (compute_rewards, nearest_vehicle_distance, MetricAccumulators) against the reference's own
evaluator code run on recorded rollouts (tests/golden/metrics.npz, oracle/gen_golden.py::gen_metrics: the unmodified
evaluators/policy_evaluator.py update_vehicle_data_dict / update_running_statistics / compute_metrics, utils/sim.py compute_reward,
evaluators/evaluator.py initialize_goal_dict / compute_nearest_dist_all, scipy's jensenshannon).  CPU only."""
import numpy as np
import pytest

from helpers import golden, cfg_of
from ctrlsim_amd import metrics, spec, scenarios
from ctrlsim_amd.evaluators.policy_evaluator import PolicyEvaluator

CASES = (("a", "a", 1, (), (0, 1, 2, 3, 4, 5, 6, 7)), ("b", "b", 5, ((1, 12), (3, 3), (7, 19)), (0, 1, 3, 4, 7, 9)),
         ("c", "c", 3, ((0, 8),), (0, 2, 5, 6, 8)))      # oracle/gen_golden.py::metrics_cases


def _inputs(tag, src, leave):
    g, cl = golden("metrics"), golden("closed_loop")
    cfg = cfg_of("loop")
    d = spec.Dims(cfg)
    rc = cl[f"{src}_recipe"]
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP,
                                  extent=float(rc[4]))
    states = cl[f"{src}_states"].copy()
    states[..., 7] = g[f"{tag}_existence"]                       # the latched existence of update_vehicle_data_dict
    accel = np.concatenate([cl[f"{src}_actions"][..., 0], np.zeros((scn.N, 1))], 1)
    return g, cfg, scn, states, cl[f"{src}_coll"], accel


@pytest.mark.parametrize("tag,src,hist,leave,evals", CASES)
def test_goal_relocation_rewards_and_nearest_distance_match_reference(tag, src, hist, leave, evals):
    g, cfg, scn, states, coll, accel = _inputs(tag, src, leave)
    gt = g[f"{tag}_gt"]
    # evaluator.py:60-76: a vehicle that leaves the log gets its last logged pose as goal
    goals = np.array([[*(gd := PolicyEvaluator.initialize_goal_dict(None, scn, v, gt[v]))["pos"], gd["heading"], gd["speed"]]
                      for v in range(scn.N)])
    np.testing.assert_allclose(goals, g[f"{tag}_goal"], rtol=0, atol=1e-12)
    if leave:
        assert np.abs(goals[leave[0][0], :2] - scn.goal_pos[leave[0][0]]).max() > 1e-3      # the goal did move
    rew = metrics.compute_rewards(states, coll, goals[:, :2], goals[:, 2], goals[:, 3], cfg.nocturne.rew_cfg)
    np.testing.assert_allclose(rew, g[f"{tag}_reward"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(metrics.nearest_vehicle_distance(states[..., :2], states[..., 7]), g[f"{tag}_nearest"], atol=1e-12)
    np.testing.assert_allclose(metrics.nearest_vehicle_distance(gt[..., :2], states[..., 7]), g[f"{tag}_gt_nearest"], atol=1e-12)


def test_running_statistics_and_metrics_match_reference():
    g = golden("metrics")
    acc = metrics.MetricAccumulators()
    for tag, src, hist, leave, evals in CASES:
        _, cfg, scn, states, coll, accel = _inputs(tag, src, leave)
        cfg = spec.make_cfg(**{**dict(dataset__waymo__max_num_agents=6, dataset__waymo__train_context_length=8,
                                      dataset__waymo__max_num_road_polylines=12, dataset__waymo__max_num_road_pts_per_polyline=10,
                                      nocturne__steps=20), "nocturne__history_steps": hist})
        goals = g[f"{tag}_goal"]
        before = {k: (acc.sums[k], acc.counts[k]) for k in acc.FIELDS}
        acc.add_scenario(states, coll, accel, g[f"{tag}_gt"][..., :5], goals[:, :2], goals[:, 2], goals[:, 3], cfg,
                         eval_ids=list(evals))
        d = lambda k: (acc.sums[k] - before[k][0], acc.counts[k] - before[k][1])
        ga, co, af = g[f"{tag}_goal_achieved"], g[f"{tag}_coll_off"], g[f"{tag}_ade_fde"]
        assert d("goal") == (ga.sum(), len(ga))
        np.testing.assert_allclose(d("coll"), (co[0].sum(), co.shape[1]), atol=1e-12)
        np.testing.assert_allclose(d("offroad"), (co[1].sum(), co.shape[1]), atol=1e-12)
        np.testing.assert_allclose(d("ade"), (af[0].sum(), af.shape[1]), atol=1e-9)
        np.testing.assert_allclose(d("fde"), (af[1].sum(), af.shape[1]), atol=1e-9)
    E = metrics.MetricAccumulators.EDGES
    w = cfg.dataset.waymo
    for key, name, lo, hi, edges in (("lin", "lin_speed", 0, 30, "lin"), ("ang", "ang_speed", -50, 50, "ang"),
                                     ("nd", "nearest_dist", 0, 40, "nd")):
        for side in ("sim", "gt"):
            ref = np.histogram(np.clip(g[f"samples_{name}_{side}"], lo, hi), bins=E[edges])[0]
            assert np.array_equal(acc.hist[f"{key}_{side}"], ref), (key, side)
    assert np.array_equal(acc.hist["accel_sim"], np.histogram(g["samples_accel_sim"], bins=E["accel"])[0])
    ga = g["samples_accel_gt"]
    ga = (np.clip(ga, w.min_accel, w.max_accel) - w.min_accel) / (w.max_accel - w.min_accel)
    ga = np.round(ga * (w.accel_discretization - 1)) / (w.accel_discretization - 1) * (w.max_accel - w.min_accel) + w.min_accel
    assert np.array_equal(acc.hist["accel_gt"], np.histogram(ga, bins=E["accel"])[0])
    m, lines = acc.compute()
    ref = dict(zip([str(k) for k in g["metric_names"]], g["metric_values"]))
    assert list(m) == list(ref)                                   # same 9 keys, same order (policy_evaluator.py:251-305)
    for k in ref:
        np.testing.assert_allclose(m[k], ref[k], rtol=1e-12, atol=1e-12, err_msg=k)
    # pack / unpack (the all-reduce payload) keeps them
    m2, _ = metrics.MetricAccumulators().unpack(acc.pack()).compute()
    assert m2 == m


INTERESTING = (("a", 14), ("b", 24), ("c", 9), ("none", 6))      # oracle/gen_golden.py::interesting_cases


@pytest.mark.parametrize("tag,N", INTERESTING)
def test_one_agent_and_two_agent_selection_match_reference(tag, N):
    """eval_mode one_agent / two_agent (cfgs/eval/base.yaml:13-14): find_interesting_agent / find_interesting_pair
    (policy_evaluator.py:308-414) pick the same vehicles from the same `random` state."""
    import random
    import types
    g = golden("interesting")
    hist, steps, rseed = (int(x) for x in g[f"{tag}_cfg"])
    ev = PolicyEvaluator.__new__(PolicyEvaluator)
    ev.cfg, ev.steps, ev.history_steps = spec.make_cfg(), steps, hist
    scn = types.SimpleNamespace(N=N, goal_pos=g[f"{tag}_goals"])
    gt = {v: {"traj": g[f"{tag}_traj"][v]} for v in range(N)}
    moving = [int(v) for v in g[f"{tag}_moving"]]
    for draw in range(6):
        random.seed(rseed + draw)
        a = ev.find_interesting_agent(scn, gt, moving)
        random.seed(rseed + draw)
        pr = ev.find_interesting_pair(scn, gt, moving)
        assert (-1 if a is None else a) == g[f"{tag}_agent"][draw]
        assert list([-1, -1] if pr is None else pr) == list(g[f"{tag}_pair"][draw])
    assert (g["none_agent"] == -1).all() and (g["b_agent"] >= 0).all()
