"""(f3) pinned: the preprocessed-dataset side of ctrlsim_amd.ingest against the reference's own dataset code run on simulated
scenes (tests/golden/preprocessed.npz, oracle/gen_golden.py::gen_preprocessed: RLWaymoDatasetCtRLSim.get_data writing the
*_physics.pkl dictionary and RLWaymoDataset.get reading it back in eval mode).  CPU only."""
import pickle

import numpy as np
import pytest

from helpers import golden, cfg_of
from ctrlsim_amd import ingest, spec, scenarios


def _export(tag):
    """The same simulated scene the fixture generator exported (oracle/gen_golden.py::export_scene), rebuilt from the recipe."""
    cl, gm = golden("closed_loop"), golden("metrics")
    cfg = cfg_of("loop")
    d = spec.Dims(cfg)
    rc = cl[f"{tag}_recipe"]
    scn = scenarios.make_scenario(int(rc[0]), int(rc[1]), n_agents=int(rc[2]), n_polylines=int(rc[3]), n_points=d.NP,
                                  extent=float(rc[4]))
    st, act, rew, ex, goals = cl[f"{tag}_states"], cl[f"{tag}_actions"], gm[f"{tag}_reward"], gm[f"{tag}_existence"], gm[f"{tag}_goal"]
    inv = {v: k for k, v in scenarios.ROAD_TYPES.items()}
    N, T1 = st.shape[:2]
    objs = [{"position": [{"x": float(st[v, t, 0]), "y": float(st[v, t, 1])} for t in range(T1)],
             "velocity": [{"x": float(st[v, t, 2]), "y": float(st[v, t, 3])} for t in range(T1)],
             "heading": [float(st[v, t, 4]) for t in range(T1)], "existence": [float(e) for e in ex[v]],
             "acceleration": [float(act[v, t, 0]) if t < T1 - 1 else 0 for t in range(T1)],
             "steering": [float(act[v, t, 1]) if t < T1 - 1 else 0 for t in range(T1)],
             "reward": [[float(x) for x in rew[v, t]] for t in range(T1)],
             "goal_position": {"x": float(goals[v, 0]), "y": float(goals[v, 1])}, "goal_heading": float(goals[v, 2]),
             "goal_speed": float(goals[v, 3]), "width": float(scn.width[v]), "length": float(scn.length[v]), "type": "vehicle"}
            for v in range(N)]
    roads = [{"geometry": [{"x": float(q[0]), "y": float(q[1])} for q in pl[:int(pl[:, 2].sum())]], "type": inv[int(np.argmax(ty))]}
             for pl, ty in zip(scn.road_points, scn.road_types)]
    return cfg, {"name": "synthetic", "objects": objs, "roads": roads}


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_preprocess_scene_and_load_preprocessed_match_reference(tag, tmp_path):
    g = golden("preprocessed")
    cfg, data = _export(tag)
    w = cfg.dataset.waymo
    pk = ingest.preprocess_scene(data, w)
    for k in ("ag_data", "ag_actions", "ag_types", "last_exist_timesteps", "ag_rewards", "ag_goals", "road_points", "road_types"):
        np.testing.assert_allclose(np.asarray(pk[k], np.float64), g[f"{tag}_pkl_{k}"], rtol=0, atol=1e-12, err_msg=k)
    for k in ("veh_edge_dist_rewards", "veh_veh_dist_rewards"):
        np.testing.assert_allclose(pk[k], g[f"{tag}_pkl_{k}"], rtol=1e-12, atol=1e-12, err_msg=k)
    assert list(pk["filtered_ag_ids"]) == list(g[f"{tag}_pkl_filtered_ag_ids"])
    # through a pickle file, as Evaluator.load_preprocessed_data reads it
    path = tmp_path / f"scene_{tag}_physics.pkl"
    with open(path, "wb") as fh:
        pickle.dump(pk, fh)
    d = ingest.load_preprocessed(str(path), w)
    assert set(d) == {"rtgs", "road_points", "road_types"}
    np.testing.assert_allclose(d["rtgs"], g[f"{tag}_rtgs"], rtol=1e-12, atol=1e-10)
    assert np.array_equal(d["road_points"], g[f"{tag}_pkl_road_points"])
    # and from the reference's own dictionary
    ref = {k[len(tag) + 5:]: g[k] for k in g.files if k.startswith(f"{tag}_pkl_")}
    np.testing.assert_allclose(ingest.load_preprocessed(ref, w)["rtgs"], g[f"{tag}_rtgs"], rtol=1e-12, atol=1e-10)


def test_json_loader_rows_match_the_reference_python_layer():
    """load_nocturne_json against get_ground_truth_states / get_road_data (utils/sim.py:20-79) run on a replay of the same
    file (tests/golden/ingest_gt.npz): trajectory rows x, y, heading, speed, existence, goal x, goal y, length for steps + 1 steps,
    existence = (x != -10000), agent-type one-hot, road lines then stop signs.  (The C++ loader itself cannot be built here.)"""
    import json
    g = golden("ingest_gt")
    data = json.loads(str(g["json"]))
    scn, info = ingest.load_nocturne_json(data, max_pts=10, steps=20)
    assert list(info["ids"]) == list(g["ids"])
    for k, i in enumerate(info["ids"]):
        np.testing.assert_allclose(info["gt_data_dict"][int(i)]["traj"], g["traj"][k], rtol=0, atol=1e-6)
        assert np.array_equal(info["gt_data_dict"][int(i)]["traj"][:, 4], g["traj"][k][:, 4])
        assert list(info["gt_data_dict"][int(i)]["type"]) == list(g["type"][k])
    rd = info["road_data"]
    assert [r["type"] for r in rd] == [str(t) for t in g["road_types"]]
    xy = np.array([[q["x"], q["y"]] for r in rd for q in ([r["geometry"]] if isinstance(r["geometry"], dict) else r["geometry"])])
    np.testing.assert_allclose(xy, g["road_xy"], rtol=0, atol=1e-12)
