"""Rollout metrics: per-step reward components and the evaluator's running statistics, on arrays.

Restates (NumPy, float64, host side — outside the timed region):
  utils/sim.py:83-141                          compute_reward: goal-reached latch within 1 m, heading/speed targets,
                                               shaped terms, vehicle-vehicle / vehicle-edge collision flags
  evaluators/evaluator.py:87-103 +
  datasets/rl_waymo/dataset.py:202-237         nearest-vehicle distance per step
  evaluators/policy_evaluator.py:162-248       update_running_statistics (goal / collision / off-road per vehicle,
                                               ADE, FDE, speed / "angular speed" / accel / nearest-dist samples)
  evaluators/policy_evaluator.py:251-305       compute_metrics: means + Jensen-Shannon distances over fixed histograms
The accumulators are plain sums and integer histograms so that ranks can be combined with ONE all-reduce
(`MetricAccumulators.pack/unpack`): this is the only collective of the multi-GPU rollout.
"""
from __future__ import annotations

import numpy as np

from .kinematics import angle_sub

N_BINS = 200
ACC_BINS = 20


def compute_rewards(states, coll, goal_pos, goal_heading, goal_speed, rew_cfg):
    """states [N,T1,8] (x,y,vx,vy,heading,len,wid,exist), coll [N,T1,2] -> reward [N,T1,8] as utils/sim.py:83-141.
    speed = |v| (Object speed_ is the norm of the body velocity, vehicle.cc:52)."""
    N, T1 = states.shape[:2]
    rew = np.zeros((N, T1, 8))
    pos = states[..., :2]
    speed = np.linalg.norm(states[..., 2:4], axis=-1)
    heading = states[..., 4]
    dist = np.linalg.norm(goal_pos[:, None, :] - pos, axis=-1)
    norm0 = dist[:, 0].copy()
    norm0[norm0 == 0.0] = 1.0
    scaling = rew_cfg.get("shaped_goal_distance_scaling", 1.0)
    rs = rew_cfg["reward_scaling"]
    achieved = np.zeros(N, bool)
    for t in range(T1):
        now = dist[:, t] < rew_cfg["position_target_tolerance"]
        rew[:, t, 0] = np.where(achieved, 1.0, now.astype(float))
        rew[:, t, 3] = np.where(achieved, scaling / rs, scaling * (1 - dist[:, t] / norm0) / rs)
        achieved = achieved | now
    rew[..., 1] = (np.abs(angle_sub(goal_heading[:, None], heading)) < rew_cfg["heading_target_tolerance"]).astype(float)
    rew[..., 2] = (np.abs(goal_speed[:, None] - speed) < rew_cfg["speed_target_tolerance"]).astype(float)
    rew[..., 4] = scaling * (1 - np.abs(speed - goal_speed[:, None]) / 40.0) / rs
    rew[..., 5] = scaling * (1 - np.abs(angle_sub(heading, goal_heading[:, None])) / (2 * np.pi)) / rs
    rew[..., 6] = coll[..., 0]
    rew[..., 7] = coll[..., 1]
    return rew


def nearest_vehicle_distance(pos, exist):
    """dataset.py:202-237 with normalize=False: pos [N,T1,2], exist [N,T1] -> [N,T1]."""
    p = pos.astype(np.float64).copy()
    p[~exist.astype(bool)] = np.inf
    with np.errstate(invalid="ignore"):
        d2 = ((p[:, None] - p[None, :]) ** 2).sum(-1)
    d2[np.isnan(d2)] = np.inf
    idx = np.arange(p.shape[0])
    d2[idx, idx] = np.inf
    d = np.sqrt(d2.min(axis=1))
    d[d == np.inf] = np.nan
    return np.nan_to_num(d * exist, nan=0.0) * exist


class MetricAccumulators:
    FIELDS = ("goal", "coll", "offroad", "ade", "fde")

    def __init__(self):
        self.sums = {k: 0.0 for k in self.FIELDS}
        self.counts = {k: 0.0 for k in self.FIELDS}
        self.hist = {k: np.zeros(N_BINS if k != "accel" else ACC_BINS, np.int64)
                     for k in ("lin_sim", "lin_gt", "ang_sim", "ang_gt", "accel_sim", "accel_gt", "nd_sim", "nd_gt")}
        self.hist["accel_sim"] = np.zeros(ACC_BINS, np.int64)
        self.hist["accel_gt"] = np.zeros(ACC_BINS, np.int64)

    # bin edges of compute_metrics (policy_evaluator.py:266,276,290,300)
    EDGES = dict(lin=np.arange(201) * 0.5 * (100 / 30), ang=np.arange(201) * 0.5 - 50,
                 accel=np.arange(ACC_BINS + 1) * 2 - ACC_BINS, nd=np.arange(201) * 0.5 * (100 / 40))

    def add_scenario(self, states, coll, applied_accel, gt_states, goal_pos, goal_heading, goal_speed, cfg,
                     eval_ids=None):
        """One finished rollout.  states [N,T1,8], coll [N,T1,2], applied_accel [N,T1] (last entry 0), gt_states
        [N,T1,5] = x, y, heading, speed, exist (synthetic scenes: scenarios.standin_log)."""
        w = cfg.dataset.waymo
        dt, hist_steps = cfg.nocturne.dt, cfg.nocturne.history_steps
        N, T1 = states.shape[:2]
        eval_ids = range(N) if eval_ids is None else eval_ids
        rew = compute_rewards(states, coll, goal_pos, goal_heading, goal_speed, cfg.nocturne.rew_cfg)
        exist = states[..., 7].astype(bool)
        nd_sim = nearest_vehicle_distance(states[..., :2], states[..., 7])
        nd_gt = nearest_vehicle_distance(gt_states[..., :2], states[..., 7])
        future = np.zeros(T1, bool)
        future[hist_steps:] = True
        colls, offs = [], []
        for v in eval_ids:
            m = exist[v] & future
            if m.sum() == 0:
                continue
            r = rew[v][m]
            self._add("goal", float(np.any(r[:, 0] == 1)))
            colls.append(float(np.any(r[:, 6] == 1)))
            offs.append(float(np.any(r[:, 7] == 1)))
            err = np.linalg.norm(states[v, :, :2] - gt_states[v, :, :2], axis=1)
            self._add("ade", err[m].mean())
            self._add("fde", err[np.where(m)[0][-1]])
            self._h("lin_sim", np.clip(np.linalg.norm(states[v, m, 2:4], axis=1), 0, 30), "lin")
            self._h("lin_gt", np.clip(gt_states[v, m, 3], 0, 30), "lin")
            self._h("ang_sim", np.clip(states[v, m, 4] / dt, -50, 50), "ang")     # "angular speed" as written, :219-220
            self._h("ang_gt", np.clip(gt_states[v, m, 2] / dt, -50, 50), "ang")
            gt_acc = np.zeros(T1)                        # central difference for 0 < t < steps - 1, else 0 (policy_evaluator.py:106-111:
            gt_acc[1:T1 - 2] = (gt_states[v, 2:T1 - 1, 3] - gt_states[v, :T1 - 3, 3]) / (2 * dt)   # steps - 1 and steps are both 0)
            am = np.ones(m.sum(), bool)
            am[0] = am[-1] = False
            ga = gt_acc[m][am]
            ga = (np.clip(ga, w.min_accel, w.max_accel) - w.min_accel) / (w.max_accel - w.min_accel)
            ga = np.round(ga * (w.accel_discretization - 1)) / (w.accel_discretization - 1)
            ga = ga * (w.max_accel - w.min_accel) + w.min_accel
            self._h("accel_gt", ga, "accel")
            self._h("accel_sim", applied_accel[v][m][am], "accel")
            self._h("nd_sim", np.clip(nd_sim[v, m], 0, 40), "nd")
            self._h("nd_gt", np.clip(nd_gt[v, m], 0, 40), "nd")
        if colls:
            self._add("coll", float(np.mean(colls)))
            self._add("offroad", float(np.mean(offs)))

    def _add(self, k, v):
        self.sums[k] += float(v)
        self.counts[k] += 1.0

    def _h(self, k, vals, edges):
        self.hist[k] += np.histogram(vals, bins=self.EDGES[edges])[0]

    # ---- one flat float64 vector for the single all-reduce
    def pack(self):
        parts = [np.array([self.sums[k] for k in self.FIELDS] + [self.counts[k] for k in self.FIELDS])]
        parts += [self.hist[k].astype(np.float64) for k in sorted(self.hist)]
        return np.concatenate(parts)

    def unpack(self, vec):
        n = len(self.FIELDS)
        for i, k in enumerate(self.FIELDS):
            self.sums[k], self.counts[k] = float(vec[i]), float(vec[n + i])
        off = 2 * n
        for k in sorted(self.hist):
            m = len(self.hist[k])
            self.hist[k] = np.rint(vec[off:off + m]).astype(np.int64)
            off += m
        return self

    def compute(self):
        """policy_evaluator.py:251-305."""
        from scipy.spatial import distance
        out = {}
        mean = lambda k: self.sums[k] / self.counts[k] if self.counts[k] else float("nan")
        out["goal"], out["collision_rate"], out["offroad_rate"] = mean("goal"), mean("coll"), mean("offroad")
        out["fde"], out["ade"] = mean("fde"), mean("ade")
        for name, a, b in (("lin_speed_jsd", "lin_sim", "lin_gt"), ("ang_speed_jsd", "ang_sim", "ang_gt"),
                           ("accel_jsd", "accel_sim", "accel_gt"), ("nearest_dist_jsd", "nd_sim", "nd_gt")):
            P, Q = self.hist[a].astype(float), self.hist[b].astype(float)
            out[name] = float(distance.jensenshannon(P / max(P.sum(), 1), Q / max(Q.sum(), 1))) if P.sum() and Q.sum() else float("nan")
        return out, ["{}: {:.6f}".format(k, v) for k, v in out.items()]
