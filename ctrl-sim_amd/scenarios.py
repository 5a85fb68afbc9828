"""Synthetic "Waymo-shaped" scenarios: deterministic functions of (base_seed, index).

No Waymo data exists in this environment, so the bench / parity configs run on random-init scenes
with the shape of a Nocturne scene as the reference's rollout sees it:
  * N vehicles with float32 pose/size (what `veh.getPosition()/getHeading()/getSpeed()` return),
    a goal (position, heading, speed)           evaluators/policy_evaluator.py:70-96
  * P_all road polylines x 100 points (x, y, exist) + one-hot types, the layout of the preprocessed
    `road_points` / `road_types`                datasets/rl_waymo/dataset.py:73-108
  * road-edge line segments (consecutive points of `road_edge` polylines), what
    Scenario::LoadRoads feeds the collision BVH  nocturne/cpp/src/scenario.cc:1006-1057
All float values are float32-exact so the device copy (fp32) and the host copy (fp64) are the same numbers.
numpy's legacy RandomState is used because its streams are frozen across numpy versions.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

ROAD_TYPES = {"none": 0, "lane": 1, "road_line": 2, "road_edge": 3, "stop_sign": 4, "crosswalk": 5,
              "speed_bump": 6, "other": 7}


@dataclass
class Scenario:
    index: int
    length: np.ndarray        # [N] f32
    width: np.ndarray         # [N] f32
    x: np.ndarray             # [N] f32
    y: np.ndarray
    heading: np.ndarray
    speed: np.ndarray
    goal_pos: np.ndarray      # [N,2] f32
    goal_heading: np.ndarray  # [N] f32
    goal_speed: np.ndarray    # [N] f32
    types: np.ndarray         # [N,5] one-hot (vehicle)
    road_points: np.ndarray   # [P_all,NP,3] f32 (x,y,exist)
    road_types: np.ndarray    # [P_all,8] one-hot
    edge_segments: np.ndarray  # [E,4] f32 (x0,y0,x1,y1)
    eval_order: np.ndarray    # [N] processing order of vehicles_to_evaluate (autoregressive_policy.py:88-94)

    @property
    def N(self):
        return len(self.x)

    def goals5(self):
        """Policy.update_state goal row (policies/policy.py:95-105), float64."""
        gh = self.goal_heading.astype(np.float64)
        gs = self.goal_speed.astype(np.float64)
        gp = self.goal_pos.astype(np.float64)
        return np.stack([gp[:, 0], gp[:, 1], gs * np.cos(gh), gs * np.sin(gh), gh], axis=1)


def _f32(a):
    return np.asarray(a, dtype=np.float32)


def make_scenario(base_seed: int, index: int, n_agents: int = 64, n_polylines: int = 512,
                  n_points: int = 100, extent: float = 100.0) -> Scenario:
    rs = np.random.RandomState((base_seed * 1000003 + index * 7919 + 12345) % (2 ** 31 - 1))
    N = n_agents
    length = _f32(rs.uniform(4.0, 5.5, N))
    width = _f32(rs.uniform(1.8, 2.3, N))
    heading = _f32(rs.uniform(-np.pi, np.pi, N))
    speed = _f32(rs.uniform(0.0, 15.0, N))
    # rejection-sample centres so that circumscribed circles are >= 1 m apart (=> box AABBs >= 1 m apart)
    xs, ys = [], []
    rad = 0.5 * np.sqrt(length.astype(np.float64) ** 2 + width.astype(np.float64) ** 2)
    for i in range(N):
        for _ in range(10000):
            px, py = rs.uniform(-extent, extent, 2)
            ok = all((px - xs[j]) ** 2 + (py - ys[j]) ** 2 > (rad[i] + rad[j] + 1.0) ** 2 for j in range(i))
            if ok:
                break
        xs.append(px)
        ys.append(py)
    x, y = _f32(xs), _f32(ys)
    gd = rs.uniform(20.0, 80.0, N)
    goal_pos = _f32(np.stack([x + gd * np.cos(heading) + rs.normal(0, 2.0, N),
                              y + gd * np.sin(heading) + rs.normal(0, 2.0, N)], 1))
    goal_heading = _f32(heading + rs.normal(0, 0.2, N))
    goal_speed = _f32(rs.uniform(0.0, 15.0, N))
    types = np.zeros((N, 5))
    types[:, 1] = 1.0

    P = n_polylines
    kinds_pool = np.array([ROAD_TYPES[k] for k in ("lane", "road_line", "road_edge", "crosswalk", "speed_bump")])
    kinds = kinds_pool[rs.randint(0, len(kinds_pool), P)]
    n_edge_min = (P + 3) // 4
    kinds[:n_edge_min] = ROAD_TYPES["road_edge"]          # >= 25 % road edges
    rs.shuffle(kinds)
    origin = rs.uniform(-1.2 * extent, 1.2 * extent, (P, 2))
    theta0 = rs.uniform(-np.pi, np.pi, P)
    curv = rs.normal(0.0, 0.02, (P, n_points))
    theta = theta0[:, None] + np.cumsum(curv, axis=1)
    pts = origin[:, None, :] + np.cumsum(np.stack([np.cos(theta), np.sin(theta)], -1), axis=1)
    npts = rs.randint(max(2, min(20, n_points // 2)), n_points + 1, P)
    exist = (np.arange(n_points)[None, :] < npts[:, None])
    rp = np.zeros((P, n_points, 3), np.float32)
    rp[..., :2] = _f32(pts) * exist[..., None]
    rp[..., 2] = exist
    rt = np.zeros((P, 8))
    rt[np.arange(P), kinds] = 1.0
    segs = []
    for p in np.where(kinds == ROAD_TYPES["road_edge"])[0]:
        q = rp[p, :npts[p], :2]
        segs.append(np.concatenate([q[:-1], q[1:]], 1))
    edge_segments = _f32(np.concatenate(segs, 0)) if segs else np.zeros((0, 4), np.float32)
    # all synthetic vehicles exist for the whole episode -> equal GT lengths; the reference's order is
    # np.argsort(lengths)[::-1] on that constant array (autoregressive_policy.py:88-94)
    eval_order = np.argsort(np.full(N, 91))[::-1].copy()
    return Scenario(index, length, width, x, y, heading, speed, goal_pos, goal_heading, goal_speed, types,
                    rp, rt, edge_segments, eval_order)


def make_batch(base_seed: int, indices, **kw):
    return [make_scenario(base_seed, int(i), **kw) for i in indices]


def standin_log(scn: Scenario, steps: int, dt: float = 0.1):
    """Stand-in for the expert log of a Waymo scene (none exist here): every vehicle drives an arc from its initial state
    with constant acceleration and a yaw rate proportional to its speed.  -> {veh: {"traj": [steps+1, 6]}} with rows
    x, y, heading, speed, exist, length — the layout `get_ground_truth_states` gives the evaluators (utils/sim.py:20-79).
    The accelerations (+-0.5 / +-1.5 m/s^2) and the curvature (steering ~ +-0.1 rad through the inverse bicycle model) sit
    in the middle of action bins: a straight constant-speed log would replay as accel = steer = 0, which is exactly a bin
    EDGE of both discretisations, so the replayed tokens would hang on float noise."""
    out = {}
    for i in range(scn.N):
        v0, L = float(scn.speed[i]), float(scn.length[i])
        a = 0.5 if i % 2 == 0 else 1.5
        if v0 >= 5.0 and i % 4 >= 2:
            a = -a
        k = 0.1003 / L * (1.0 if (i // 2) % 2 == 0 else -1.0)
        tr = np.zeros((steps + 1, 6))
        x, y, h, v = float(scn.x[i]), float(scn.y[i]), float(scn.heading[i]), v0
        for t in range(steps + 1):
            tr[t] = (x, y, h, v, 1.0, L)
            x += v * np.cos(h) * dt
            y += v * np.sin(h) * dt
            h += k * v * dt
            v = max(v + a * dt, 0.0)
        out[i] = {"traj": tr}
    return out
