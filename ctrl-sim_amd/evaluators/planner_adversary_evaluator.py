"""`PlannerAdversaryEvaluator(cfg, planner, adversary).evaluate_planner_adversary() -> (metrics_dict, [str])` — the second
in-repo caller of the policy plugin surface (reference: evaluators/planner_adversary_evaluator.py:26-593, driver
eval_planner.py:16-137).

Two policies share one simulated scene: the planner (cfgs/policy/ctrl_sim_planner.yaml: tilts +10/+10/+10) drives the ego
vehicle, the adversary (ctrl_sim_adversary.yaml: vehicle-vehicle tilt -10) drives one other vehicle, everybody else — and
those two before `history_steps - 1` — replays the log through the inverse bicycle model.  Each policy keeps its own
history buffers, key names (`next_planner_acceleration`, `planner_rtgs`, ...) and device session, exactly as in the
reference loop (:497-546); an adversary named "cat" is not a policy but a fixed trajectory (`apply_adv_traj`, :163-199).

Forced by the environment (as in PolicyEvaluator): scenes come from `cfg.eval_planner_adversary.synthetic`, the log is the
stand-in of scenarios.standin_log, and — there being no CAT dictionary (cfg.cat.dict_path) — the ego is
vehicle 0 and the adversary the vehicle nearest to it at t = 0.  Pinned by tests/golden/planner_adversary.npz (two
unmodified reference policies + the real FreeCar/Box2D)."""
from __future__ import annotations

import random

import numpy as np
import torch
from scipy.spatial import distance

from .. import scenarios as _scn
from ..kinematics import bicycle_backward
from ..simulation import Simulation
from .policy_evaluator import PolicyEvaluator

PLANNER_KEYS = {"next_acceleration": "next_planner_acceleration", "next_steering": "next_planner_steering",
                "rtgs": "planner_rtgs"}                                       # eval_planner.py:19-23
ADVERSARY_KEYS = {"next_acceleration": "next_adversary_acceleration", "next_steering": "next_adversary_steering",
                  "rtgs": "adversary_rtgs"}                                   # eval_planner.py:76-80


def pick_ego_adversary(scn):
    """Stand-in of the CAT dictionary lookup (planner_adversary_evaluator.py:431-456)."""
    d = np.hypot(scn.x - scn.x[0], scn.y - scn.y[0])
    d[0] = np.inf
    return 0, int(np.argmin(d))


class PlannerAdversaryEvaluator(PolicyEvaluator):
    def __init__(self, cfg, planner, adversary):
        self.planner = planner
        self.adversary = adversary
        self.cfg_pa = cfg.eval_planner_adversary
        super().__init__(cfg, planner)
        self.history_steps = self.cfg_pa.history_steps
        syn = self.cfg_pa.get("synthetic") or dict(num_scenarios=1, n_agents=8, n_polylines=12, seed=0, extent=25.0)
        self.synthetic = dict(syn)
        self.ego_vehicle = None
        self.adversary_vehicle = None

    # ---- planner_adversary_evaluator.py:48-76
    def reset(self):
        seed = self.cfg.eval_planner_adversary.seed
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        self.ades_all, self.fdes_all, self.goal_achieved_all, self.progress_all = [], [], [], []
        self.collision_rate_scenario, self.collision_rate_w_adv_scenario, self.offroad_rate_scenario = [], [], []
        self.ego_jerk_all, self.ego_steering_rate_all, self.ego_accel_all = [], [], []
        self.lin_speed_sim_all, self.lin_speed_gt_all, self.ang_speed_sim_all, self.ang_speed_gt_all = [], [], [], []
        self.accel_sim_all, self.accel_gt_all, self.nearest_dist_sim_all, self.nearest_dist_gt_all = [], [], [], []
        self.collision_speed_with_ego = []

    # ---- :79-108
    def initialize_vehicle_data_dict(self, veh, goal_dict):
        d = super().initialize_vehicle_data_dict(veh, goal_dict)
        for k in ("rtgs", "next_acceleration", "next_steering"):
            d.pop(k)
        d.update({"planner_rtgs": [], "next_planner_acceleration": 0., "next_planner_steering": 0., "adversary_rtgs": [],
                  "next_adversary_acceleration": 0., "next_adversary_steering": 0.})
        return d

    # ---- :163-199
    def apply_adv_traj(self, veh, t, gt_data_dict, vdd, adv_traj):
        v = veh.getID()
        traj = gt_data_dict[v]["traj"]
        exists = traj[t][4] and traj[t + 1][4]
        if t > 0 and vdd[v]["existence"][-1] == 0:
            exists = 0
        if not exists:
            a, s = 0.0, 0.0
            veh.setPosition(-1000000, -1000000)
        else:
            nxt = np.array([[adv_traj[t + 1, 0], adv_traj[t + 1, 1], adv_traj[t + 1, 4],
                             np.sqrt(adv_traj[t + 1, 2] ** 2 + adv_traj[t + 1, 3] ** 2), traj[t + 1][-1]]])
            prev = np.array([[veh.getPosition().x, veh.getPosition().y, veh.getHeading(), veh.getSpeed()]])
            a, s = bicycle_backward(nxt, prev, self.dt)
            a, s = float(a[0]), float(s[0])
        if a > 0.0:
            veh.acceleration = a
        else:
            veh.brake(np.abs(a))
        veh.steering = s
        return veh, [a, s]

    # ---- :458-546
    def evaluate_planner_adversary(self, adv_traj_fn=None):
        """adv_traj_fn(scn, gt_data_dict, ego, adv) -> [steps+1, 5] (x, y, vx, vy, yaw): the fixed adversarial trajectory used
        when adversary.name == 'cat' (the reference reads it from the CAT dictionary)."""
        self.reset()
        syn = self.synthetic
        d_model = self.planner.model.dims
        n_eval = 0
        for k in range(int(syn["num_scenarios"])):
            if n_eval == self.cfg_pa.num_files_to_evaluate:
                break
            scn = _scn.make_scenario(int(syn.get("seed", 0)), k, n_agents=int(syn["n_agents"]),
                                     n_polylines=int(syn["n_polylines"]), n_points=d_model.NP,
                                     extent=float(syn.get("extent", 100.0)))
            self.planner.scenario_index = self.adversary.scenario_index = scn.index
            gt_data_dict = self._ground_truth(scn)
            sim = Simulation(scn, device=self.planner.model.device, steps=self.steps, dt=self.dt)
            vehicles = sim.getScenario().vehicles()
            for veh in vehicles:
                veh.expert_control = False
                veh.physics_simulated = True
            self.ego_vehicle, self.adversary_vehicle = pick_ego_adversary(scn)
            cat = self.adversary.name == "cat"
            adv_traj = adv_traj_fn(scn, gt_data_dict, self.ego_vehicle, self.adversary_vehicle) if cat else None
            n_eval += 1
            preproc_data = {"road_points": scn.road_points.astype(np.float64), "road_types": scn.road_types.copy()}
            vdd, goal_dict, goal_norm = {}, {}, {}
            for veh in vehicles:
                v = veh.getID()
                goal_dict[v] = {"pos": scn.goal_pos[v].astype(np.float64), "heading": float(scn.goal_heading[v]),
                                "speed": float(scn.goal_speed[v])}
                vdd[v] = self.initialize_vehicle_data_dict(veh, goal_dict[v])
                goal_norm[v] = np.linalg.norm(np.array([veh.getPosition().x, veh.getPosition().y]) - goal_dict[v]["pos"])
            self.planner.reset(vdd)
            if not cat:
                self.adversary.reset(vdd)
            for t in range(self.steps):
                vdd = self.update_vehicle_data_dict(t, vehicles, vdd, goal_dict, goal_norm, gt_data_dict)
                self.planner.update_state(vdd, [self.ego_vehicle], t)
                if not cat:
                    self.adversary.update_state(vdd, [self.adversary_vehicle], t)
                vdd = self.planner.predict(vdd, gt_data_dict, preproc_data, None, [self.ego_vehicle], t)
                if not cat:
                    vdd = self.adversary.predict(vdd, gt_data_dict, preproc_data, None, [self.adversary_vehicle], t)
                for veh in vehicles:
                    v = veh.getID()
                    if t >= self.history_steps - 1 and v == self.ego_vehicle:
                        veh, act = self.planner.act(veh, t, vdd)
                    elif t >= self.history_steps - 1 and v == self.adversary_vehicle:
                        if cat:
                            veh, act = self.apply_adv_traj(veh, t, gt_data_dict, vdd, adv_traj)
                        else:
                            veh, act = self.adversary.act(veh, t, vdd)
                    else:
                        veh, act = self.apply_gt_action(veh, t, gt_data_dict, vdd)
                    vdd[v]["acceleration"].append(act[0])
                    vdd[v]["steering"].append(act[1])
                sim.step(self.dt)
            vdd = self.update_vehicle_data_dict(self.steps, vehicles, vdd, goal_dict, goal_norm, gt_data_dict)
            for veh in vehicles:
                vdd[veh.getID()]["acceleration"].append(0)
                vdd[veh.getID()]["steering"].append(0)
            self.last_vehicle_data_dict = vdd
            self.update_running_statistics(vdd)
        return self.compute_metrics()

    # ---- :202-365
    def update_running_statistics(self, data_dict):
        T1, hs = self.steps + 1, self.history_steps
        future = np.zeros(T1, bool)
        future[hs:] = True
        xy = lambda v, key: np.array([[p["x"], p["y"]] for p in data_dict[v][key]])
        collisions, collisions_w_adv, offroads = [], [], []
        has_adv_ego_collision = False

        v = self.ego_vehicle
        ego_mask = np.array(data_dict[v]["existence"]).astype(bool) * future
        if ego_mask.sum() != 0:
            rew = np.array(data_dict[v]["reward"])[ego_mask]
            goal_achieved = np.any(np.sum(rew[:, :1], axis=1) == 1)
            self.goal_achieved_all.append(float(goal_achieved))
            collisions.append(float(np.any(rew[:, 6] == 1)))
            offroads.append(float(np.any(rew[:, 7] == 1)))
            sim_pos, gt_pos = xy(v, "position"), xy(v, "gt_position")
            self.ades_all.append(np.linalg.norm(sim_pos[ego_mask] - gt_pos[ego_mask], axis=1).mean())
            last = np.where(ego_mask == 1)[-1][-1]
            self.fdes_all.append(np.linalg.norm(sim_pos[last] - gt_pos[last]))
            seg = sim_pos[hs:last + 1]
            step_len = np.linalg.norm(np.diff(seg, axis=0), axis=-1)
            if goal_achieved:
                progress = step_len.sum()
            else:                                                      # only the steps that bring the ego closer to its goal
                dist_to_goal = np.linalg.norm(seg - gt_pos[last][None], axis=-1)
                progress = step_len[np.diff(dist_to_goal) < 0].sum()
            self.progress_all.append(progress)
            acc = np.array(data_dict[v]["acceleration"])[ego_mask]
            self.ego_jerk_all.append(np.abs(np.diff(acc)) / self.dt)
            self.ego_accel_all.append(np.abs(acc))
            self.ego_steering_rate_all.append(np.abs(np.diff(np.array(data_dict[v]["steering"])[ego_mask])) / self.dt)

        v = self.adversary_vehicle
        adv_mask = np.array(data_dict[v]["existence"]).astype(bool) * future
        if adv_mask.sum() != 0:
            vel = xy(v, "velocity")[adv_mask]
            self.lin_speed_sim_all.append(np.linalg.norm(vel, axis=1)[:, None])
            self.lin_speed_gt_all.append(np.array(data_dict[v]["gt_speed"])[adv_mask][:, None])
            self.ang_speed_sim_all.append((np.array(data_dict[v]["heading"])[adv_mask] / self.dt)[:, None])
            self.ang_speed_gt_all.append((np.array(data_dict[v]["gt_heading"])[adv_mask] / self.dt)[:, None])
            gt_acc = np.array(data_dict[v]["gt_acceleration"])[adv_mask]
            sim_acc = np.array(data_dict[v]["acceleration"])[adv_mask]
            inner = np.ones(gt_acc.shape, bool)                        # central differences exist only inside the window
            inner[0] = inner[-1] = False
            self.accel_sim_all.append(sim_acc[inner][:, None])
            self.accel_gt_all.append(gt_acc[inner][:, None])
            self.nearest_dist_gt_all.append(np.array(data_dict[v]["gt_nearest_dist"])[adv_mask][:, None])
            self.nearest_dist_sim_all.append(np.array(data_dict[v]["nearest_dist"])[adv_mask][:, None])

        e, a = self.ego_vehicle, self.adversary_vehicle
        if ego_mask.sum() != 0 and adv_mask.sum() != 0:
            ego_coll = np.array(data_dict[e]["reward"])[ego_mask, 6]
            adv_coll = np.array(data_dict[a]["reward"])[adv_mask, 6]
            n = min(len(ego_coll), len(adv_coll))
            ego_coll, adv_coll = ego_coll[:n], adv_coll[:n]
            both = ((ego_coll == adv_coll).astype(float) * ego_coll).astype(bool)
            has = float(np.any(both))
            if has == 1.:                                              # both flagged: is it each other?  (distance test)
                ego_pos, adv_pos = xy(e, "position")[ego_mask][:n], xy(a, "position")[adv_mask][:n]
                adv_v = xy(a, "velocity")
                valid = False
                for c in np.where(both)[0]:
                    if np.linalg.norm(ego_pos[c] - adv_pos[c]) < data_dict[e]["length"] + data_dict[a]["length"]:
                        valid = True
                        # the reference appends sqrt(vx[c]^2 + vy^2) with vy the WHOLE array (:345); its mean is what
                        # compute_metrics reports, so the same array is kept
                        self.collision_speed_with_ego.append(np.sqrt(adv_v[c, 0] ** 2 + adv_v[:, 1] ** 2))
                        break
                if not valid:
                    has = 0.
            collisions_w_adv.append(has)
            has_adv_ego_collision = bool(has)
        if len(collisions) > 0:
            self.collision_rate_scenario.append(np.array(collisions).mean())
            if len(collisions_w_adv) == 0:
                collisions_w_adv.append(0.)
            self.collision_rate_w_adv_scenario.append(np.array(collisions_w_adv).mean())
            self.offroad_rate_scenario.append(np.array(offroads).mean())
        return has_adv_ego_collision

    # ---- :368-428
    def compute_metrics(self):
        w = self.cfg_rl_waymo
        m = {}
        m["ego_goal"] = np.array(self.goal_achieved_all).mean()
        m["ego_prog"] = np.array(self.progress_all).mean()
        m["ego_cr"] = np.array(self.collision_rate_scenario).mean()
        m["ego_cr_w_adv"] = np.array(self.collision_rate_w_adv_scenario).mean()
        m["ego_or"] = np.array(self.offroad_rate_scenario).mean()
        m["ego_fde"] = np.array(self.fdes_all).mean()
        m["ego_ade"] = np.array(self.ades_all).mean()
        m["ego_accel"] = np.concatenate(self.ego_accel_all, axis=0).mean()
        m["ego_jerk"] = np.concatenate(self.ego_jerk_all, axis=0).mean()
        m["ego_steer_rate"] = np.concatenate(self.ego_steering_rate_all, axis=0).mean()
        m["adv_coll_speed"] = np.array(self.collision_speed_with_ego).mean() if self.collision_speed_with_ego else float("nan")

        def jsd(sim, gt, lo, hi, edges):
            sim = np.clip(np.concatenate(sim, axis=0), lo, hi) if lo is not None else np.concatenate(sim, axis=0)
            gt = np.clip(np.concatenate(gt, axis=0), lo, hi) if lo is not None else np.concatenate(gt, axis=0)
            P = np.histogram(sim, bins=edges)[0] / len(sim)
            Q = np.histogram(gt, bins=edges)[0] / len(gt)
            return distance.jensenshannon(P, Q)
        m["adv_lin_jsd"] = jsd(self.lin_speed_sim_all, self.lin_speed_gt_all, 0, 30, np.arange(201) * 0.5 * (100 / 30))
        m["adv_ang_jsd"] = jsd(self.ang_speed_sim_all, self.ang_speed_gt_all, -50, 50, np.arange(201) * 0.5 - 50)
        gt_acc = np.concatenate(self.accel_gt_all, axis=0)              # discretise / undiscretise the log's accelerations
        gt_acc = (np.clip(gt_acc, a_min=w.min_accel, a_max=w.max_accel) - w.min_accel) / (w.max_accel - w.min_accel)
        gt_acc = np.round(gt_acc * (w.accel_discretization - 1)) / (w.accel_discretization - 1)
        gt_acc = gt_acc * (w.max_accel - w.min_accel) + w.min_accel
        m["adv_acc_jsd"] = jsd(self.accel_sim_all, [gt_acc], None, None,
                               np.arange(w.accel_discretization + 1) * 2 - w.accel_discretization)
        m["nearest_dist_jsd"] = jsd(self.nearest_dist_sim_all, self.nearest_dist_gt_all, 0, 40,
                                    np.arange(201) * 0.5 * (100 / 40))
        return m, ["{}: {:.6f}".format(k, v) for (k, v) in m.items()]
