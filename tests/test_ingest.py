"""Scenario ingest (ctrlsim_amd/ingest.py): road chunking pinned against the reference's own RLWaymoDataset.get_roads
(tests/golden/ingest.npz), the Nocturne JSON reader checked by write / read round trips and against the loader rules of
nocturne/cpp/src/scenario.cc:893-1057."""
import json

import numpy as np

from helpers import golden
from ctrlsim_amd import ingest, scenarios


def test_road_chunking_matches_reference_get_roads():
    g = golden("ingest")
    road_data = json.loads(bytes(g["road_json"]).decode())
    pts, types, edges = ingest.roads_to_polylines(road_data, 100)
    assert pts.shape == g["road_points"].shape == (12, 100, 3)        # 250 -> 3 chunks, 100 -> exactly 1, 101 -> 2, 200 -> 2
    assert np.array_equal(pts, g["road_points"]) and np.array_equal(types, g["road_types"])
    assert len(edges) == int(g["n_edges"])
    for i, e in enumerate(edges):
        assert np.array_equal(e, g[f"edge{i}"])
    # empty road list
    p0, t0, e0 = ingest.roads_to_polylines([], 100)
    assert p0.shape == (0, 100, 3) and t0.shape == (0, 8) and e0 == []


def _scene():
    scn = scenarios.make_scenario(3, 0, n_agents=6, n_polylines=7, n_points=100, extent=40.0)
    log = scenarios.standin_log(scn, 20)
    log[2]["traj"][12:, 4] = 0                                       # vehicle 2 leaves the log at step 12
    log[4]["traj"][:3, 4] = 0                                        # vehicle 4 is not there at the start: never spawned
    return scn, log


def test_nocturne_json_round_trip_and_loader_rules():
    scn, log = _scene()
    d = ingest.scenario_to_nocturne_json(scn, log)
    d["objects"].insert(1, dict(d["objects"][0], type="pedestrian"))   # consumes id 1, is not spawned
    d["roads"].append({"geometry": [{"x": 1.5, "y": -2.5}], "type": "stop_sign"})
    d["objects"][0]["heading"][0] = float(np.rad2deg(scn.heading[0])) + 720.0     # degrees, any winding
    s2, info = ingest.load_nocturne_json(json.loads(json.dumps(d)), steps=20)
    keep = [0, 1, 2, 3, 5]                                           # vehicle 4 dropped
    assert list(info["ids"]) == [0, 2, 3, 4, 5]                      # pedestrian took id 1; dropped vehicle took none
    assert s2.N == 5 and np.array_equal(s2.types, np.eye(5)[[1] * 5])
    for k in ("x", "y", "length", "width", "speed"):
        assert np.array_equal(getattr(s2, k), getattr(scn, k)[keep]), k
    np.testing.assert_allclose(s2.heading, scn.heading[keep], atol=1e-6, rtol=0)
    assert np.all(np.abs(s2.heading) <= np.pi)
    assert np.array_equal(s2.goal_pos, scn.goal_pos[keep])
    # goal heading / speed = those of the last VALID step (scenario.cc:943-947)
    tr2 = log[2]["traj"]
    np.testing.assert_allclose(s2.goal_speed[2], tr2[11, 3], rtol=1e-6)
    # polylines: identical chunks, the stop sign appended as one repeated point of type 4
    assert np.array_equal(s2.road_points[:7], scn.road_points) and np.array_equal(s2.road_types[:7], scn.road_types)
    assert np.array_equal(s2.road_points[7], np.tile(np.float32([1.5, -2.5, 1.0]), (100, 1))) and s2.road_types[7, 4] == 1
    assert np.array_equal(s2.edge_segments, scn.edge_segments)       # road-edge point pairs = collision segments
    # ground truth rows as get_ground_truth_states gives them
    gt = info["gt_data_dict"]
    t3 = gt[3]["traj"]                                               # = vehicle 2 of the source scene
    assert t3.shape == (21, 8) and t3[11, 4] == 1 and t3[12, 4] == 0 and t3[12, 0] == -10000
    np.testing.assert_allclose(t3[:12, :2], tr2[:12, :2].astype(np.float32), rtol=0, atol=0)
    assert t3[0, 7] == scn.length[2] and np.array_equal(t3[0, 5:7], scn.goal_pos[2])
    assert gt[0]["type"] == [0.0, 1.0, 0.0]
    # processing order: decreasing number of existing log steps (autoregressive_policy.py:88-94)
    assert int(s2.eval_order[-1]) == 2
    assert info["moving"].all()


def test_parked_vehicle_is_not_moving_and_short_log_is_padded_invalid():
    scn, log = _scene()
    d = ingest.scenario_to_nocturne_json(scn, log)
    o = d["objects"][5]
    p0 = o["position"][0]
    o["position"] = [dict(p0) for _ in o["position"]]
    o["velocity"] = [{"x": 0.0, "y": 0.0} for _ in o["velocity"]]
    o["goalPosition"] = dict(p0)
    _, info = ingest.load_nocturne_json(d, steps=30)                 # asks for more steps than the file holds
    assert list(info["moving"]) == [True, True, True, True, False]
    tr = info["gt_data_dict"][4]["traj"]                             # source vehicle 5 (vehicle 4 took no id)
    assert tr.shape == (31, 8) and tr[20, 4] == 1 and tr[21, 4] == 0 and tr[30, 0] == -10000
