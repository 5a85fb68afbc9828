"""Real-time (dense) rewards of the rollout driver — what feeds the RTGs of policies with `real_time_rewards = True`
(the Decision-Transformer baseline, cfgs/policy/dt.yaml).  Host-side float64 NumPy, one call per simulator step.

Reference chain: Evaluator.compute_dense_reward (evaluators/evaluator.py:106-140) ->
RLWaymoDataset.compute_dist_to_nearest_road_edge_rewards / compute_dist_to_nearest_vehicle_rewards / compute_rewards
(datasets/rl_waymo/dataset.py:187-275) -> compute_distance_to_road_edge and the signed distance to polylines
(utils/data.py:152-290, itself the Waymo Open Sim Agents metric).  Pinned by tests/golden/dense_reward.npz (the reference's own
functions on random scenes)."""
from __future__ import annotations

import numpy as np

_CYCLIC_TOLERANCE_M2 = 1.0          # utils/data.py:17


def _cross(a, b):
    return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]


def _dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]


def signed_distance_to_polyline(xys, polyline):
    """utils/data.py:215-290: negative on the port side (inside the road boundary), positive on the starboard side."""
    polyline = np.asarray(polyline, np.float64)
    is_cyclic = np.square(polyline[0] - polyline[-1]).sum() < _CYCLIC_TOLERANCE_M2
    starts, ends = polyline[None, :-1, :2], polyline[None, 1:, :2]
    s2p = xys[:, None, :2] - starts
    s2e = ends - starts
    with np.errstate(invalid="ignore", divide="ignore"):
        rel_t = np.nan_to_num(_dot(s2p, s2e) / _dot(s2e, s2e))
    n = np.sign(_cross(s2p, s2e))
    dist = np.linalg.norm(s2p - s2e * np.clip(rel_t, 0.0, 1.0)[..., None], axis=-1)
    pad = np.concatenate([s2e[:, -1:], s2e, s2e[:, :1]], axis=1)
    convex = _cross(pad[:, :-1], pad[:, 1:]) > 0.0
    n_prior = np.concatenate([np.where(is_cyclic, n[:, -1:], n[:, :1]), n[:, :-1]], axis=-1)
    n_next = np.concatenate([n[:, 1:], np.where(is_cyclic, n[:, :1], n[:, -1:])], axis=-1)
    before = np.where(convex[:, :-1], np.maximum(n, n_prior), np.minimum(n, n_prior))
    after = np.where(convex[:, 1:], np.maximum(n, n_next), np.minimum(n, n_next))
    sign = np.where(rel_t < 0.0, before, np.where(rel_t < 1.0, n, after))
    k = np.argmin(dist, axis=-1)[:, None]
    return np.take_along_axis(sign, k, axis=1)[:, 0] * np.min(dist, axis=-1)


def signed_distance_to_road_edges(xy, road_edge_polylines):
    """compute_distance_to_road_edge (utils/data.py:152-181): the signed distance to the NEAREST polyline (by |distance|)."""
    d = [signed_distance_to_polyline(xy, p) for p in road_edge_polylines if len(p) >= 2]
    d = np.stack(d, axis=-1)
    return np.take_along_axis(d, np.argmin(np.abs(d), axis=-1)[:, None], axis=1)[:, 0]


def nearest_vehicle_distance_raw(xy, exist):
    """compute_dist_to_nearest_vehicle_rewards(normalize=False) at one step: 0 for a vehicle that does not exist or is alone."""
    pos = np.array(xy, np.float64)
    ex = np.asarray(exist, bool)
    pos[~ex] = np.inf
    with np.errstate(invalid="ignore"):
        sq = np.sum((pos[:, None] - pos[None, :]) ** 2, axis=-1)
    np.fill_diagonal(sq, np.inf)
    with np.errstate(invalid="ignore"):
        d = np.sqrt(np.min(sq, axis=1))
    d[d == np.inf] = np.nan
    return np.nan_to_num(d * ex.astype(np.float64), nan=0.0)


def dense_reward(xy, exist, rewards, road_edge_polylines, w):
    """Evaluator.compute_dense_reward for one step.  xy [N,2] positions, exist [N], rewards [N,8] = one compute_reward row per
    vehicle (utils/sim.py:83-141), w = cfg.dataset.waymo.  -> (dense [N,3] = goal, vehicle, road-edge components; nearest [N]
    raw distance to the nearest existing vehicle).
    NOTE which row the reference uses: it stacks every reward row recorded so far, broadcasts the current distances against
    them and then takes `all_rewards[i, 0]` (evaluator.py:135-138) — the flags of STEP 0, not of the current step.  The caller
    reproduces that by passing the step-0 rows."""
    ex = np.asarray(exist, np.float64)
    r = np.asarray(rewards, np.float64) * ex[:, None]
    edge = -signed_distance_to_road_edges(np.asarray(xy, np.float64), road_edge_polylines) / w.dist_to_road_edge_scaling_factor
    edge = edge * ex
    nearest = nearest_vehicle_distance_raw(xy, ex) * ex
    veh = np.clip(nearest, 0.0, w.max_veh_veh_distance) / w.max_veh_veh_distance
    # compute_rewards (dataset.py:239-275) — existence = column 2 of the [x, y, exist] rows handed over by the evaluator
    if w.remove_shaped_goal:
        goal = r[:, 0] * w.pos_target_achieved_rew_multiplier
    else:
        goal = r[:, 0] * w.pos_target_achieved_rew_multiplier + \
            (np.clip(r[:, 3], w.pos_goal_shaped_min, w.pos_goal_shaped_max) - w.pos_goal_shaped_max) / w.pos_goal_shaped_max
    if w.remove_shaped_veh_reward:
        vv = -r[:, 6] * w.veh_veh_collision_rew_multiplier
    else:
        vv = veh - r[:, 6] * w.veh_veh_collision_rew_multiplier
    if w.remove_shaped_edge_reward:
        ve = -r[:, 7] * w.veh_edge_collision_rew_multiplier
    else:
        ve = np.clip(np.abs(edge) * w.dist_to_road_edge_scaling_factor, 0, 5) / 5.0 - r[:, 7] * w.veh_edge_collision_rew_multiplier
    dense = np.stack([goal, vv, ve], 1) * ex[:, None]
    return dense, nearest * 1.0


def normalize_rtgs(rtgs, w):
    """AutoregressivePolicy.get_data:73-78: clip to the configured range and scale to [0, 1], per component."""
    out = np.array(rtgs, np.float64)
    for c, (lo, hi) in enumerate(((w.min_rtg_pos, w.max_rtg_pos), (w.min_rtg_veh, w.max_rtg_veh),
                                  (w.min_rtg_road, w.max_rtg_road))):
        out[..., c] = (np.clip(out[..., c], lo, hi) - lo) / (hi - lo)
    return out
