"""Scenario ingest: Nocturne-format Waymo JSON -> the arrays the rollout path works on.

What the reference does on the way from a scenario file to model inputs, restated on the host in NumPy:

* `Scenario::LoadObjects` / `LoadRoads` (nocturne/cpp/src/scenario.cc:893-1057): vehicles valid at the start time with
  length / width / pose / speed = |velocity| / goal = `goalPosition`, target heading and speed = those of the last valid
  step, ids counted over every spawnable object (non-vehicles consume an id even when they are not allowed in), headings
  given in degrees and normalised to [-pi, pi], expert trajectories kept for log replay; road-edge polylines become the
  collision segments (consecutive point pairs), everything in float32 as in C++.
* `get_ground_truth_states` (utils/sim.py:20-59): stepping the expert-controlled scenario = reading the expert arrays;
  rows `x, y, heading, speed, existence, goal x, goal y, length` with existence = (x != -10000).
* `get_road_data` (utils/sim.py:61-74) + `RLWaymoDataset.get_roads` (datasets/rl_waymo/dataset.py:73-108): polylines cut
  into chunks of `max_num_road_pts_per_polyline` points with an existence channel, the last chunk zero-padded, a stop sign =
  its position repeated along the chunk, one-hot road types (utils/data.py:334-337).

* the preprocessed dataset the evaluators read map features and initial RTGs from (`Evaluator.load_preprocessed_data`,
  evaluators/evaluator.py:44-57 -> `RLWaymoDataset.get`, datasets/rl_waymo/dataset.py:458-500 -> `RLWaymoDatasetCtRLSim.get_data`,
  dataset_ctrl_sim.py:38-97): `preprocess_scene` builds the `*_physics.pkl` dictionary from a simulated scene in the evaluators'
  export format (`extract_rawdata`, the road-edge / nearest-vehicle distance rewards: dataset.py:111-237), `load_preprocessed`
  turns such a dictionary (or pickle file) into `{'rtgs', 'road_points', 'road_types'}` (`compute_rewards` + reverse cumulative
  sum: dataset.py:240-275, dataset_ctrl_sim.py:92-97).

`roads_to_polylines`, `preprocess_scene` and `load_preprocessed` are pinned against the reference's own dataset code run on
synthetic scenes (tests/golden/ingest.npz, preprocessed.npz).  The Nocturne JSON object loader has no reference binary to run
against here (Scenario::LoadObjects needs SFML to build): it follows the C++ line by line, is checked by a write / read round
trip, and its output is pinned only as far as the reference's PYTHON layer goes — `get_ground_truth_states` / `get_road_data`
(utils/sim.py:20-79) run on a replay of the same file give the same rows (tests/golden/ingest_gt.npz).
`scenario_to_nocturne_json` is the inverse (exports a synthetic scene in the same format)."""
from __future__ import annotations

import json

import numpy as np

from .scenarios import ROAD_TYPES, Scenario

OBJECT_TYPES = ("unset", "vehicle", "pedestrian", "cyclist", "other")      # object_base.h ObjectType
INVALID_POSITION = -10000.0                                                # the file format's marker of a missing step


def _heading_from_degrees(deg):
    """NormalizeAngle(Radians(float(deg))) as scenario.cc:934-935 evaluates it (geometry_utils.h:38-58): float argument,
    double arithmetic against the double constants, float results; range [-pi, pi]."""
    rad = np.float32(np.float64(np.float32(deg)) / 180.0 * np.pi)
    ret = np.float32(np.fmod(np.float64(rad), 2.0 * np.pi))
    if np.float64(ret) > np.pi:
        return np.float32(np.float64(ret) - 2.0 * np.pi)
    if np.float64(ret) < -np.pi:
        return np.float32(np.float64(ret) + 2.0 * np.pi)
    return ret


def road_type_onehot(name):
    return np.eye(len(ROAD_TYPES))[ROAD_TYPES[name]]


def road_data_from_json(roads_json):
    """get_road_data(scenario): [{"geometry": [{"x","y"}...] | {"x","y"} (stop sign), "type": str}] with the coordinates
    as the C++ side holds them (float32).  Road lines first, then stop signs (utils/sim.py:63-72)."""
    f = lambda v: float(np.float32(v))
    lines, stops = [], []
    for road in roads_json:
        geo = road["geometry"]
        if road["type"] == "stop_sign":
            stops.append({"geometry": {"x": f(geo[0]["x"]), "y": f(geo[0]["y"])}, "type": "stop_sign"})
        else:
            name = road["type"] if road["type"] in ROAD_TYPES else "other"
            lines.append({"geometry": [{"x": f(p["x"]), "y": f(p["y"])} for p in geo], "type": name})
    return lines + stops


def roads_to_polylines(roads_data, max_pts=100):
    """RLWaymoDataset.get_roads (datasets/rl_waymo/dataset.py:73-108) -> (road_points [P, max_pts, 3] float64,
    road_types [P, 8] one-hot, road_edge_polylines [list of [n, 2]])."""
    pts, types, edges = [], [], []
    for road in roads_data:
        geo = road["geometry"]
        if isinstance(geo, dict):                            # stop sign: the point repeated along the chunk
            pts.append(np.tile(np.array([geo["x"], geo["y"], 1.0]), (max_pts, 1)))
            types.append(road_type_onehot(road["type"]))
            continue
        xy = np.array([[p["x"], p["y"]] for p in geo], np.float64).reshape(-1, 2)
        if road["type"] == "road_edge":
            edges.append(xy.copy())
        for c0 in range(0, len(xy), max_pts):                # full chunks, then a zero-padded remainder
            chunk = np.zeros((max_pts, 3))
            n = min(max_pts, len(xy) - c0)
            chunk[:n, :2] = xy[c0:c0 + n]
            chunk[:n, 2] = 1.0
            pts.append(chunk)
            types.append(road_type_onehot(road["type"]))
    if not pts:
        return np.zeros((0, max_pts, 3)), np.zeros((0, len(ROAD_TYPES))), edges
    return np.array(pts), np.array(types), edges


def load_nocturne_json(src, index=0, max_pts=100, start_time=0, allow_non_vehicles=False, spawn_invalid_objects=False,
                       steps=90, moving_threshold=0.2, speed_threshold=0.05):
    """-> (Scenario, info) from a Nocturne scenario file (path, file object or the parsed dict).
    info: ids [N] (the simulator's object ids), moving [N] bool (getObjectsThatMoved), gt_data_dict {id: {"traj": [steps+1, 8],
    "type": one-hot}} as get_ground_truth_states returns it, road_data (get_road_data), road_edge_polylines."""
    if isinstance(src, dict):
        data = src
    elif hasattr(src, "read"):
        data = json.load(src)
    else:
        with open(src) as fh:
            data = json.load(fh)
    f32 = np.float32
    rows, ids, moving, gt = [], [], [], {}
    cur_id = 0
    for obj in data["objects"]:
        kind = obj["type"] if obj["type"] in OBJECT_TYPES else "other"
        pos = np.array([[p["x"], p["y"]] for p in obj["position"]], f32)
        vel = np.array([[p["x"], p["y"]] for p in obj["velocity"]], f32)
        heading = np.array([_heading_from_degrees(h) for h in obj["heading"]], f32)
        speed = np.sqrt(vel[:, 0] * vel[:, 0] + vel[:, 1] * vel[:, 1]).astype(f32)
        valid = np.array([bool(v) for v in obj["valid"]])
        goal = np.array([obj["goalPosition"]["x"], obj["goalPosition"]["y"]], f32) if "goalPosition" in obj else np.zeros(2, f32)
        if not valid[start_time] and not spawn_invalid_objects:
            continue                                         # not there at the start: no object, no id
        last = np.where(valid)[0]
        tgt_heading = heading[last[-1]] if len(last) else f32(0)
        tgt_speed = speed[last[-1]] if len(last) else f32(0)
        is_moving = bool(np.any(valid & ((speed > speed_threshold) |
                                         (np.hypot(pos[:, 0] - goal[0], pos[:, 1] - goal[1]) > moving_threshold))))
        if kind == "vehicle" or (allow_non_vehicles and kind in ("pedestrian", "cyclist")):
            rows.append(dict(length=f32(obj["length"]), width=f32(obj["width"]), x=pos[start_time, 0], y=pos[start_time, 1],
                             heading=heading[start_time], speed=speed[start_time], goal=goal, goal_heading=tgt_heading,
                             goal_speed=tgt_speed, kind=kind))
            ids.append(cur_id)
            moving.append(is_moving)
            T1 = steps + 1
            tr = np.zeros((T1, 8))
            n = min(T1, len(pos) - start_time)
            sl = slice(start_time, start_time + n)
            tr[:n, 0], tr[:n, 1], tr[:n, 2], tr[:n, 3] = pos[sl, 0], pos[sl, 1], heading[sl], speed[sl]
            tr[:n, 4] = (pos[sl, 0] != f32(INVALID_POSITION)).astype(np.float64)
            tr[n:, 0] = tr[n:, 1] = INVALID_POSITION
            tr[:, 5], tr[:, 6], tr[:, 7] = goal[0], goal[1], f32(obj["length"])
            # get_agent_type_onehot(veh.getType().value) = np.eye(3)[value], value = 1 vehicle, 2 pedestrian (utils/data.py:326-328;
            # a cyclist, value 3, is out of range there — zeros here)
            # log_len = rows of the file's log from start_time on: the rows behind it are padding, and the reference's
            # expert_trajectories_.at(id).at(current_time_) (scenario.cc:280) throws there
            gt[cur_id] = {"traj": tr, "log_len": int(len(pos) - start_time),
                          "type": [float(kind == "unset"), float(kind == "vehicle"), float(kind == "pedestrian")]}
        cur_id += 1                                          # every spawnable object consumes an id (scenario.cc:992-997)

    road_data = road_data_from_json(data.get("roads", []))
    road_points, road_types, edge_polys = roads_to_polylines(road_data, max_pts)
    segs = [np.concatenate([p[:-1], p[1:]], 1) for p in edge_polys if len(p) > 1]
    edge_segments = np.concatenate(segs).astype(f32) if segs else np.zeros((0, 4), f32)
    N = len(rows)
    col = lambda k: np.array([r[k] for r in rows], f32).reshape(N, *np.shape(rows[0][k])) if N else np.zeros((0,), f32)
    types = np.zeros((N, 5))
    for i, r in enumerate(rows):
        types[i, OBJECT_TYPES.index(r["kind"])] = 1.0
    lengths = [int(gt[i]["traj"][:, 4].sum()) for i in ids]
    eval_order = np.argsort(np.array(lengths))[::-1].copy() if N else np.zeros(0, np.int64)
    scn = Scenario(index=index, length=col("length"), width=col("width"), x=col("x"), y=col("y"), heading=col("heading"),
                   speed=col("speed"), goal_pos=col("goal").reshape(N, 2), goal_heading=col("goal_heading"),
                   goal_speed=col("goal_speed"), types=types, road_points=road_points.astype(f32), road_types=road_types,
                   edge_segments=edge_segments, eval_order=eval_order)
    scn.road_edge_polylines = edge_polys                     # unchunked, for the real-time road-edge distance reward
    info = dict(ids=np.array(ids, np.int64), moving=np.array(moving, bool), gt_data_dict=gt, road_data=road_data,
                road_edge_polylines=edge_polys, name=data.get("name", ""))
    return scn, info


def scenario_to_nocturne_json(scn: Scenario, log, name="synthetic"):
    """Scenario + expert log {veh: {"traj": rows x, y, heading, speed, exist, ...}} -> a dict in the Nocturne file format
    (objects with per-step position / velocity / heading in DEGREES / valid, goalPosition, type; roads with point lists)."""
    objects = []
    for i in range(scn.N):
        tr = np.asarray(log[i]["traj"], np.float64)
        ex = tr[:, 4] > 0
        x = np.where(ex, tr[:, 0], INVALID_POSITION)
        y = np.where(ex, tr[:, 1], INVALID_POSITION)
        kind = OBJECT_TYPES[int(np.argmax(scn.types[i]))] if scn.types.shape[1] == 5 else "vehicle"
        objects.append({"position": [{"x": float(a), "y": float(b)} for a, b in zip(x, y)],
                        "width": float(scn.width[i]), "length": float(scn.length[i]),
                        "heading": [float(np.rad2deg(h)) for h in tr[:, 2]],
                        "velocity": [{"x": float(s * np.cos(h)), "y": float(s * np.sin(h))} for s, h in zip(tr[:, 3], tr[:, 2])],
                        "valid": [bool(e) for e in ex],
                        "goalPosition": {"x": float(scn.goal_pos[i, 0]), "y": float(scn.goal_pos[i, 1])}, "type": kind})
    inv = {v: k for k, v in ROAD_TYPES.items()}
    roads = []
    for pl, ty in zip(scn.road_points, scn.road_types):
        n = int(pl[:, 2].sum())
        roads.append({"geometry": [{"x": float(p[0]), "y": float(p[1])} for p in pl[:n]], "type": inv[int(np.argmax(ty))]})
    return {"name": name, "objects": objects, "roads": roads, "tl_states": {}}


# ------------------------------------------------------------------------------------------------ preprocessed dataset (*_physics.pkl)
def object_type_onehot(name):
    """utils/data.py get_object_type_onehot: one-hot over (unset, vehicle, pedestrian, cyclist, other)."""
    return np.eye(len(OBJECT_TYPES))[OBJECT_TYPES.index(name if name in OBJECT_TYPES else "other")]


def preprocess_scene(data, w, idx=0):
    """The dictionary `RLWaymoDatasetCtRLSim.get_data` pickles for one simulated scene (dataset_ctrl_sim.py:54-90): `data` is the
    evaluators' export {"objects": [{position, velocity, heading, existence, acceleration, steering, reward, goal_position,
    goal_heading, goal_speed, length, width, type}], "roads": get_road_data(...)}; w = cfg.dataset.waymo."""
    from .rewards import signed_distance_to_road_edges
    from .metrics import nearest_vehicle_distance
    objs = data["objects"]
    road_points, road_types, edge_polys = roads_to_polylines(data["roads"], w.max_num_road_pts_per_polyline)
    ag_data, ag_actions, ag_rewards, ag_types, ag_goals, incomplete, last_exist = [], [], [], [], [], [], []
    for n, o in enumerate(objs):                                  # extract_rawdata, dataset.py:111-186
        pos = np.array([[p["x"], p["y"]] for p in o["position"]], np.float64)
        vel = np.array([[p["x"], p["y"]] for p in o["velocity"]], np.float64)
        heading = np.array(o["heading"], np.float64).reshape(-1, 1)
        ex = np.array(o["existence"], np.float64).reshape(-1, 1)
        gone = np.where(ex == 0.0)[0]
        if len(gone) > 0:
            assert np.all(ex[gone[0]:] == 0.0), "existence must not come back"
        if len(gone) > 0 and gone[0] == 0:
            incomplete.append(n)
            last_exist.append(-1)
        else:
            last_exist.append(int(np.where(ex == 1.0)[0][-1]))
        T = len(pos)
        gh, gs = o["goal_heading"], o["goal_speed"]
        goal = np.array([o["goal_position"]["x"], o["goal_position"]["y"], gs * np.cos(gh), gs * np.sin(gh), gh])
        ag_data.append(np.concatenate([pos, vel, heading, np.ones((T, 1)) * o["length"], np.ones((T, 1)) * o["width"], ex], -1))
        ag_actions.append(np.column_stack((o["acceleration"], o["steering"])))
        ag_rewards.append(np.array(o["reward"], np.float64) * ex)
        ag_types.append(object_type_onehot(o["type"]))
        ag_goals.append(np.repeat(goal[None], T, 0))
    ag_data = np.array(ag_data)
    # distance rewards (dataset.py:187-237), zeroed where the vehicle does not exist
    edge = np.array([-signed_distance_to_road_edges(ag_data[n, :, :2], edge_polys) / w.dist_to_road_edge_scaling_factor
                     for n in range(len(objs))]) * ag_data[:, :, -1]
    nd = nearest_vehicle_distance(ag_data[:, :, :2], ag_data[:, :, -1])
    veh = np.nan_to_num(np.clip(nd, 0.0, w.max_veh_veh_distance) / w.max_veh_veh_distance) * ag_data[:, :, -1]
    return dict(idx=idx, num_agents=len(objs), road_points=road_points, road_types=road_types, ag_data=ag_data,
                ag_actions=np.array(ag_actions), ag_types=np.array(ag_types), last_exist_timesteps=np.array(last_exist),
                veh_edge_dist_rewards=edge, veh_veh_dist_rewards=veh, ag_rewards=np.array(ag_rewards),
                filtered_ag_ids=[i for i in range(len(objs)) if i not in incomplete], ag_goals=np.array(ag_goals))


def load_preprocessed(src, w):
    """`Evaluator.load_preprocessed_data` -> `RLWaymoDataset.get` in eval mode: the preprocessed dictionary (or the path of its
    pickle) -> {'rtgs' [N,T,5] returns-to-go per reward component (goal position, heading, speed, vehicle, road edge),
    'road_points', 'road_types'} (dataset.py:240-275, 493-498; dataset_ctrl_sim.py:92-97)."""
    if not isinstance(src, dict):
        import pickle
        with open(src, "rb") as fh:
            src = pickle.load(fh)
    ag_data = np.asarray(src["ag_data"], np.float64)
    r = np.asarray(src["ag_rewards"], np.float64)
    ex = ag_data[:, :, -1:]
    edge, veh = np.asarray(src["veh_edge_dist_rewards"]), np.asarray(src["veh_veh_dist_rewards"])
    if w.remove_shaped_goal:
        goal = r[:, :, 0] * w.pos_target_achieved_rew_multiplier
    else:
        goal = r[:, :, 0] * w.pos_target_achieved_rew_multiplier + \
            (np.clip(r[:, :, 3], w.pos_goal_shaped_min, w.pos_goal_shaped_max) - w.pos_goal_shaped_max) * (1 / w.pos_goal_shaped_max)
    head = r[:, :, 1] + r[:, :, 5]
    speed = r[:, :, 2] + r[:, :, 4]
    vv = (-1 * r[:, :, 6] * w.veh_veh_collision_rew_multiplier) if w.remove_shaped_veh_reward else \
        (veh - r[:, :, 6] * w.veh_veh_collision_rew_multiplier)
    ve = (-1 * r[:, :, 7] * w.veh_edge_collision_rew_multiplier) if w.remove_shaped_edge_reward else \
        (np.clip(np.abs(edge) * w.dist_to_road_edge_scaling_factor, 0, 5) / 5. - r[:, :, 7] * w.veh_edge_collision_rew_multiplier)
    allr = np.concatenate([x[:, :, None] * ex for x in (goal, head, speed, vv, ve)], -1)
    rtgs = np.cumsum(allr[:, ::-1], axis=1)[:, ::-1]
    return {"rtgs": rtgs, "road_points": src["road_points"], "road_types": src["road_types"]}
