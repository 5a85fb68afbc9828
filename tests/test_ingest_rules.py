"""The C++ half of the scenario loader (nocturne/cpp/src/scenario.cc:893-1057: Scenario::LoadObjects / LoadRoads), as a TABLE of
hand-written cases with hand-derived expectations — no repo writer in the loop, every expectation cites the source line it
restates.  (nocturne_core needs SFML and cannot be built here; the Python half of the loader is pinned against the reference's
own code by tests/test_ingest_pinned.py.)"""
import numpy as np
import pytest

from ctrlsim_amd import ingest

T = 4                                      # steps in the hand-written file


def _obj(kind, xs, ys, valid, heading_deg, vx, vy, goal=None, length=4.5, width=2.0):
    o = {"type": kind, "width": width, "length": length,
         "position": [{"x": x, "y": y} for x, y in zip(xs, ys)],
         "velocity": [{"x": a, "y": b} for a, b in zip(vx, vy)],
         "heading": list(heading_deg), "valid": list(valid)}
    if goal is not None:
        o["goalPosition"] = {"x": goal[0], "y": goal[1]}
    return o


FILE = {
    "name": "hand_written",
    "objects": [
        # 0: vehicle, valid throughout, moving (speed 5 > 0.05)                         -> spawned, id 0, moving
        _obj("vehicle", [0, 1, 2, 3], [0, 0, 0, 0], [1, 1, 1, 1], [0, 0, 0, 0], [5, 5, 5, 5], [0, 0, 0, 0], goal=(30, 0)),
        # 1: vehicle, INVALID at start_time 0                                            -> skipped, consumes NO id (:957-961)
        _obj("vehicle", [-10000, 5, 6, 7], [-10000, 5, 5, 5], [0, 1, 1, 1], [90] * 4, [0, 1, 1, 1], [0] * 4, goal=(9, 5)),
        # 2: pedestrian, valid                                                           -> not stored unless allow_non_vehicles (:973),
        #                                                                                   but ++cur_id runs (:997): consumes id 1
        _obj("pedestrian", [2, 2, 2, 2], [8, 8, 8, 8], [1, 1, 1, 1], [45] * 4, [0] * 4, [0] * 4, goal=(2, 8)),
        # 3: vehicle, valid at 0 and 1, gone afterwards; heading 450 deg = 90 deg       -> id 2; target heading / speed = LAST VALID step
        #                                                                                   (:943-947): heading(1) = 135 deg, |v(1)| = 5
        _obj("vehicle", [10, 10, -10000, -10000], [0, 3, -10000, -10000], [1, 1, 0, 0], [450, 135, 0, 0], [0, 3, 0, 0], [3, 4, 0, 0],
             goal=(10, 40), length=5.0, width=2.2),
        # 4: vehicle, parked AT its goal (distance 0.1 <= 0.2) with speed 0.01 <= 0.05 -> id 3, NOT in moving_objects (:948-951)
        _obj("vehicle", [20, 20, 20, 20], [20, 20, 20, 20], [1, 1, 1, 1], [-190] * 4, [0.01] * 4, [0] * 4, goal=(20.1, 20)),
        # 5: vehicle without goalPosition: target_position stays (0, 0) (:903-907)      -> id 4, moving (distance to (0,0) > 0.2)
        _obj("vehicle", [7, 7, 7, 7], [-7, -7, -7, -7], [1, 1, 1, 1], [180] * 4, [0] * 4, [0] * 4),
    ],
    "roads": [
        {"type": "road_edge", "geometry": [{"x": 0, "y": -5}, {"x": 10, "y": -5}, {"x": 20, "y": -6}]},   # 2 collision segments (:1029-1035)
        {"type": "lane", "geometry": [{"x": 0, "y": 0}, {"x": 50, "y": 0}]},                              # a road line, no segments
        {"type": "stop_sign", "geometry": [{"x": 3, "y": 4}, {"x": 99, "y": 99}]},                        # first point only (:1011-1014)
        {"type": "road_edge", "geometry": [{"x": 5, "y": 5}]},                                             # one point: a line, NO segment
    ],
    "tl_states": {},
}


def test_default_config_table():
    scn, info = ingest.load_nocturne_json(FILE, start_time=0, allow_non_vehicles=False, spawn_invalid_objects=False, steps=T - 1,
                                          moving_threshold=0.2, speed_threshold=0.05)
    assert list(info["ids"]) == [0, 2, 3, 4]                     # :894,957-961,997 (object 1 skipped before ++cur_id; the pedestrian took 1)
    assert list(info["moving"]) == [True, True, False, True]     # :948-953
    assert scn.N == 4
    np.testing.assert_array_equal(scn.x, np.float32([0, 10, 20, 7]))             # position at current_time_ (:899-900)
    np.testing.assert_array_equal(scn.speed, np.float32([5, 3, 0.01, 0]))        # |velocity| at current_time_ (:936-937,965)
    np.testing.assert_array_equal(scn.length, np.float32([4.5, 5.0, 4.5, 4.5]))
    # headings: degrees -> radians, normalised to [-pi, pi] (:934-935): 450 -> pi/2, -190 -> +170 deg; 180 -> MINUS pi, because
    # Radians() returns float(pi) = 3.14159274 and NormalizeAngle compares it with the double constant kPi = 3.14159265...
    # (geometry_utils.h:54-58: ret > kPi -> ret - kTwoPi)
    np.testing.assert_allclose(scn.heading, np.float32([0.0, np.pi / 2, np.deg2rad(170.0), -np.pi]), atol=1e-6)
    # targets: position = goalPosition or (0, 0) (:903-907); heading / speed of the last valid step (:943-947)
    np.testing.assert_array_equal(scn.goal_pos, np.float32([[30, 0], [10, 40], [20.1, 20], [0, 0]]))
    np.testing.assert_allclose(scn.goal_heading, np.float32([0.0, np.deg2rad(135.0), np.deg2rad(170.0), -np.pi]), atol=1e-6)
    np.testing.assert_allclose(scn.goal_speed, np.float32([5.0, 5.0, 0.01, 0.0]), atol=1e-6)
    # get_ground_truth_states rows (utils/sim.py:23-38): existence = (x != -10000), rows x, y, heading, speed, exist, goal x, goal y, length
    tr = info["gt_data_dict"][2]["traj"]
    assert tr.shape == (T, 8)
    np.testing.assert_array_equal(tr[:, 4], [1, 1, 0, 0])
    np.testing.assert_array_equal(tr[2:, 0], [-10000, -10000])
    np.testing.assert_allclose(tr[1, :4], [10, 3, np.deg2rad(135.0), 5.0], atol=1e-6)
    np.testing.assert_array_equal(tr[:, 5:8], np.tile([10, 40, 5.0], (T, 1)))
    # roads: two road lines + the one-point edge as polylines (chunks of 100 points), then the stop sign (get_road_data: lines, then
    # stop signs, utils/sim.py:63-72); collision segments only from consecutive road-edge points (:1029-1035)
    assert [r["type"] for r in info["road_data"]] == ["road_edge", "lane", "road_edge", "stop_sign"]
    assert info["road_data"][3]["geometry"] == {"x": 3.0, "y": 4.0}
    np.testing.assert_array_equal(scn.edge_segments, np.float32([[0, -5, 10, -5], [10, -5, 20, -6]]))
    assert scn.road_points.shape == (4, 100, 3) and scn.road_points[0, :, 2].sum() == 3 and scn.road_points[2, :, 2].sum() == 1
    np.testing.assert_array_equal(scn.road_points[3], np.tile(np.float32([3, 4, 1]), (100, 1)))        # dataset.py:84-90


@pytest.mark.parametrize("start_time,allow,spawn,ids,kinds", [
    (0, True, False, [0, 1, 2, 3, 4], ["vehicle", "pedestrian", "vehicle", "vehicle", "vehicle"]),   # pedestrian spawned with its id (:974-983)
    (0, False, True, [0, 1, 3, 4, 5], ["vehicle"] * 5),        # spawn_invalid_objects: object 1 is spawned at its (invalid) position, id 1
    (2, False, False, [0, 1, 3, 4], ["vehicle"] * 4),          # start_time 2: object 1 valid (id 1), the pedestrian takes 2, object 3 INVALID -> skipped
])
def test_config_switches_table(start_time, allow, spawn, ids, kinds):
    scn, info = ingest.load_nocturne_json(FILE, start_time=start_time, allow_non_vehicles=allow, spawn_invalid_objects=spawn,
                                          steps=T - 1)
    assert list(info["ids"]) == ids
    assert [ingest.OBJECT_TYPES[int(np.argmax(t))] for t in scn.types] == kinds
    if start_time == 2:
        np.testing.assert_array_equal(scn.x, np.float32([2, 6, 20, 7]))          # position at current_time_ = 2
    if spawn:
        assert scn.x[1] == np.float32(-10000)                  # spawned where the file says it is: the invalid marker
