"""`PolicyEvaluator(cfg, policy).evaluate_policy() -> (metrics_dict, [str])` — the rollout driver of the plugin surface
(reference: evaluators/policy_evaluator.py:27-595, evaluators/evaluator.py:24-193, utils/sim.py:83-141).

The per-scenario loop is the reference's (update_vehicle_data_dict -> policy.update_state -> policy.predict ->
policy.act / apply_gt_action -> sim.step -> update_running_statistics -> compute_metrics) with the same dict schema
(policy_evaluator.py:70-96), so a `Policy` written against the reference runs here unchanged.  Differences, all forced
by the environment: scenarios come from `cfg.eval.synthetic` (no Nocturne JSON / preprocessed pickles exist here) and
"ground truth" is a stand-in log (constant-acceleration arcs from the initial state, scenarios.standin_log); the simulator is
`ctrlsim_amd.simulation.Simulation` (HIP kernels) instead of the pybind `nocturne_cpp` module."""
from __future__ import annotations

import random

import numpy as np
import torch

from .. import scenarios as _scn
from ..kinematics import bicycle_backward, angle_sub
from ..metrics import MetricAccumulators
from ..simulation import Simulation, CollisionType


class PolicyEvaluator:
    def __init__(self, cfg, policy):
        self.cfg = cfg
        self.cfg_rl_waymo = cfg.dataset.waymo
        self.steps = cfg.nocturne.steps
        self.dt = cfg.nocturne.dt
        self.history_steps = cfg.nocturne.history_steps
        self.policy = policy
        self.vehicles_to_evaluate = None
        syn = cfg.eval.get("synthetic") or dict(num_scenarios=1, n_agents=8, n_polylines=40, seed=0, extent=60.0)
        self.synthetic = dict(syn)
        self.reset()

    def reset(self):
        seed = self.cfg.eval.seed
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        self.acc = MetricAccumulators()

    # ---- policy_evaluator.py:70-96
    def initialize_vehicle_data_dict(self, veh, goal_dict):
        return {"gt_position": [], "gt_speed": [], "gt_heading": [], "gt_acceleration": [], "gt_nearest_dist": [],
                "position": [], "velocity": [], "heading": [], "nearest_dist": [], "existence": [], "acceleration": [],
                "steering": [], "reward": [], "dense_reward": [],
                "goal_position": {"x": goal_dict["pos"][0], "y": goal_dict["pos"][1]},
                "goal_heading": goal_dict["heading"], "goal_speed": goal_dict["speed"], "width": veh.getWidth(),
                "length": veh.getLength(), "type": "vehicle", "timestep": [], "rtgs": [], "next_acceleration": 0.,
                "next_steering": 0.}

    # ---- utils/sim.py:83-141
    def compute_reward(self, veh, goal, goal_dist_normalizer, d):
        rew_cfg = self.cfg.nocturne.rew_cfg
        pos = np.array([veh.position.x, veh.position.y])
        prev = len(d["reward"]) > 0 and d["reward"][-1][0]
        pos_t = float(True) if prev else float(np.linalg.norm(goal["pos"] - pos) < rew_cfg["position_target_tolerance"])
        spd_t = float(np.abs(goal["speed"] - veh.speed) < rew_cfg["speed_target_tolerance"])
        hd_t = float(np.abs(angle_sub(goal["heading"], veh.heading)) < rew_cfg["heading_target_tolerance"])
        sc, rs = rew_cfg.get("shaped_goal_distance_scaling", 1.0), rew_cfg["reward_scaling"]
        norm = goal_dist_normalizer if goal_dist_normalizer != 0.0 else 1.0
        pos_r = sc / rs if prev else sc * (1 - np.linalg.norm(goal["pos"] - pos) / norm) / rs
        spd_r = sc * (1 - np.abs(veh.speed - goal["speed"]) / 40.0) / rs
        hd_r = sc * (1 - np.abs(angle_sub(veh.heading, goal["heading"])) / (2 * np.pi)) / rs
        return [pos_t, hd_t, spd_t, pos_r, spd_r, hd_r, float(veh.collision_type_veh == CollisionType.VEHICLE_VEHICLE),
                float(veh.collision_type_edge == CollisionType.VEHICLE_ROAD)]

    # ---- policy_evaluator.py:99-159 (real_time_rewards=False branch)
    def update_vehicle_data_dict(self, t, vehicles, vdd, goal_dict, goal_norm, gt_data_dict, preproc_data=None):
        for veh_idx, veh in enumerate(vehicles):
            v = veh.getID()
            gt = np.array(gt_data_dict[v]["traj"])
            d = vdd[v]
            d["gt_position"].append({"x": gt[t, 0], "y": gt[t, 1]})
            d["gt_heading"].append(gt[t, 2])
            d["gt_speed"].append(gt[t, 3])
            d["gt_acceleration"].append((gt[t + 1, 3] - gt[t - 1, 3]) / (2 * self.dt) if 0 < t < self.steps - 1 else 0)
            d["position"].append({"x": veh.getPosition().x, "y": veh.getPosition().y})
            d["velocity"].append({"x": veh.velocity().x, "y": veh.velocity().y})
            d["heading"].append(veh.getHeading())
            d["timestep"].append(t)
            ex = gt[t, 4]
            if t > 0 and d["existence"][-1] == 0:
                ex = 0
            d["existence"].append(ex)
            if self.policy.real_time_rewards:                      # policy_evaluator.py:122-147: the RTG the policy is fed
                key = self.policy.key_dict["rtgs"]
                if t == 0:
                    # policy_evaluator.py:123-146: the logged return-to-go of the preprocessed dataset — components (goal position,
                    # heading, speed, vehicle, road edge) -> (goal position, vehicle, road edge) — unless the policy asks for the
                    # maximum / minimum achievable return
                    if preproc_data is not None and "rtgs" in preproc_data:
                        r5 = np.asarray(preproc_data["rtgs"][veh_idx, t], np.float64)
                        rtg = np.concatenate([r5[:1], r5[3:]], axis=-1)
                    elif self.policy.max_return or self.policy.min_return:
                        rtg = np.zeros(3)
                    else:
                        raise ValueError("real_time_rewards without max_return / min_return starts from the preprocessed dataset's "
                                         "RTGs: pass cfg.eval.preprocessed_files (ctrlsim_amd.ingest.load_preprocessed)")
                    if self.policy.max_return or self.policy.min_return:
                        rtg[:] = (10.0, 90.0, 90.0)                # the maximum achievable return
                    if self.policy.min_return and not self.policy.max_return and v in self.vehicles_to_evaluate:
                        rtg[:] = (0.0, -10.0, -10.0)               # evaluated vehicles: the minimum possible return
                    d[key].append(rtg)
                else:
                    d[key].append(d[key][-1] - d["dense_reward"][-1])
            d["reward"].append(self.compute_reward(veh, goal_dict[v], goal_norm[v], d))
        if self.policy.real_time_rewards:
            return self.compute_dense_reward(t, vdd)
        ids = list(vdd.keys())
        pos = np.array([[vdd[v]["position"][t]["x"], vdd[v]["position"][t]["y"]] for v in ids])
        gpos = np.array([[vdd[v]["gt_position"][t]["x"], vdd[v]["gt_position"][t]["y"]] for v in ids])
        ex = np.array([vdd[v]["existence"][t] for v in ids], float)
        from ..metrics import nearest_vehicle_distance
        nd = nearest_vehicle_distance(pos[:, None], ex[:, None])[:, 0]
        gnd = nearest_vehicle_distance(gpos[:, None], ex[:, None])[:, 0]
        for i, v in enumerate(ids):
            vdd[v]["nearest_dist"].append(nd[i])
            vdd[v]["gt_nearest_dist"].append(gnd[i])
        return vdd

    # ---- evaluators/evaluator.py:106-140
    def compute_dense_reward(self, t, vdd):
        from ..rewards import dense_reward, nearest_vehicle_distance_raw
        ids = list(vdd.keys())
        xy = np.array([[vdd[v]["position"][t]["x"], vdd[v]["position"][t]["y"]] for v in ids])
        gxy = np.array([[vdd[v]["gt_position"][t]["x"], vdd[v]["gt_position"][t]["y"]] for v in ids])
        ex = np.array([vdd[v]["existence"][t] for v in ids], float)
        step0 = np.array([vdd[v]["reward"][0] for v in ids], float)   # the reference indexes its reward stack at step 0 (:137)
        dense, nearest = dense_reward(xy, ex, step0, self.road_edge_polylines, self.cfg_rl_waymo)
        gt_nearest = nearest_vehicle_distance_raw(gxy, ex) * ex
        scale = self.cfg_rl_waymo.max_veh_veh_distance                 # as written there (:126-127): metres times 15
        for i, v in enumerate(ids):
            vdd[v]["nearest_dist"].append(nearest[i] * scale)
            vdd[v]["gt_nearest_dist"].append(gt_nearest[i] * scale)
            vdd[v]["dense_reward"].append(dense[i])
        return vdd

    @staticmethod
    def extract_road_edge_polylines(scn):
        """evaluators/evaluator.py:143-157 on the scenario arrays: the road_edge polylines as point lists."""
        polys = getattr(scn, "road_edge_polylines", None)
        if polys is not None:
            return polys
        out = []
        for pl, ty in zip(scn.road_points, scn.road_types):
            if int(np.argmax(ty)) == 3:
                out.append(np.asarray(pl[:int(pl[:, 2].sum()), :2], np.float64))
        return out

    # ---- evaluators/evaluator.py:160-193
    def apply_gt_action(self, veh, t, gt_data_dict, vdd):
        v = veh.getID()
        traj = gt_data_dict[v]["traj"]
        exists = traj[t][4] and traj[t + 1][4]
        if t > 0 and vdd[v]["existence"][-1] == 0:
            exists = 0
        if not exists:
            a, s = 0.0, 0.0
            veh.setPosition(-1000000, -1000000)
        else:
            nxt = np.array([[traj[t + 1][0], traj[t + 1][1], traj[t + 1][2], traj[t + 1][3], traj[t + 1][-1]]])
            prev = np.array([[veh.getPosition().x, veh.getPosition().y, veh.getHeading(), veh.getSpeed()]])
            a, s = bicycle_backward(nxt, prev, self.dt)
            a, s = float(a[0]), float(s[0])
        if a > 0.0:
            veh.acceleration = a
        else:
            veh.brake(np.abs(a))
        veh.steering = s
        return veh, [a, s]

    def _ground_truth(self, scn):
        """The stand-in expert log (scenarios.standin_log): traj rows = x, y, heading, speed, exist, length."""
        return _scn.standin_log(scn, self.steps, self.dt)

    def _scenes(self, syn):
        """-> (scenario, ground-truth dict keyed by vehicle index, indices of the vehicles that move): Nocturne JSON files
        when cfg.eval.scenario_files lists any (ctrlsim_amd.ingest: the reference's load_scenario + get_ground_truth_states +
        get_moving_vehicles + preprocessed road arrays), synthetic scenes otherwise."""
        d_model = self.policy.model.dims
        files = self.cfg.eval.get("scenario_files")
        if files:
            from .. import ingest
            pre_files = self.cfg.eval.get("preprocessed_files") or [None] * len(files)
            for k, path in enumerate(files):
                scn, info = ingest.load_nocturne_json(path, index=k, max_pts=d_model.NP, steps=self.steps)
                gt = {i: info["gt_data_dict"][int(vid)] for i, vid in enumerate(info["ids"])}
                # Evaluator.load_preprocessed_data (evaluators/evaluator.py:44-57): the scene's *_physics.pkl, if there is one
                pre = ingest.load_preprocessed(pre_files[k], self.cfg_rl_waymo) if pre_files[k] is not None else None
                yield scn, gt, [i for i in range(scn.N) if info["moving"][i]], pre
            return
        for k in range(int(syn["num_scenarios"])):
            scn = _scn.make_scenario(int(syn.get("seed", 0)), k, n_agents=int(syn["n_agents"]),
                                     n_polylines=int(syn["n_polylines"]), n_points=d_model.NP,
                                     extent=float(syn.get("extent", 100.0)))
            yield scn, self._ground_truth(scn), list(range(scn.N)), None

    # ---- evaluators/evaluator.py:60-76
    def initialize_goal_dict(self, scn, v, gt_traj):
        pos, heading, speed = scn.goal_pos[v].astype(np.float64), float(scn.goal_heading[v]), float(scn.goal_speed[v])
        gone = np.where(gt_traj[:, 4] == 0)[0]
        if len(gone) > 0:                                              # leaves the log: the goal is where it was last seen
            i = gone[0] - 1
            if np.linalg.norm(gt_traj[i, :2] - pos) > 0.0:
                pos, heading, speed = gt_traj[i, :2], gt_traj[i, 2], gt_traj[i, 3]
        return {"pos": pos, "heading": heading, "speed": speed}

    # ---- policy_evaluator.py:308-414
    def find_interesting_pair(self, scn, gt_data_dict, moving):
        """An ordered pair of moving vehicles whose goals are close in space and time, drawn with `random.choice` from every such
        pair in row-major order (policy_evaluator.py:362-414); find_interesting_agent (:308-359) is its first element and draws
        the same random number.  Goals are the relocated ones (last logged position of a vehicle that leaves the log, :326-331),
        both trajectories must hold at least interesting_traj_len_threshold logged steps after the history.  None: no such pair."""
        e = self.cfg.eval
        ids = [v for v in range(scn.N) if v in moving]                 # `vehicles` order, moving ones only (:319-322)
        if not ids:
            return None
        goals = np.zeros((len(ids), 2))
        goal_t = np.zeros(len(ids), np.int64)
        long_enough = np.zeros(len(ids), bool)
        for k, v in enumerate(ids):
            traj = np.array(gt_data_dict[v]["traj"])
            pos, i_goal = scn.goal_pos[v].astype(np.float64), self.steps - 1
            gone = np.where(traj[:, 4] == 0)[0]
            if len(gone) > 0:
                i_goal = gone[0] - 1
                if np.linalg.norm(traj[i_goal, :2] - pos) > 0.0:
                    pos = traj[i_goal, :2]
            goals[k], goal_t[k] = pos, i_goal - self.history_steps
            long_enough[k] = traj[self.history_steps:, 4].sum() >= e.interesting_traj_len_threshold
        dist = np.linalg.norm(goals[None] - goals[:, None], 2, -1)
        ok = (dist < e.interesting_goal_dist_threshold) & (dist > 0) & long_enough[:, None] & long_enough[None, :] & \
             (np.abs(goal_t[:, None] - goal_t[None, :]) < e.interesting_timestep_diff_threshold)
        pairs = list(zip(*np.where(ok)))
        if not pairs:
            return None
        a, b = random.choice(pairs)
        return [ids[a], ids[b]]

    def find_interesting_agent(self, scn, gt_data_dict, moving):
        pair = self.find_interesting_pair(scn, gt_data_dict, moving)
        return None if pair is None else pair[0]

    def _choose_vehicles(self, scn, gt_data_dict, moving):
        """policy_evaluator.py:448-466 — the only consumer of `random`: models seeded alike evaluate the same vehicles."""
        mode = self.cfg.eval.eval_mode
        if mode == "multi_agent":
            thr = self.cfg.eval.multi_agent_eval_threshold
            return random.sample(moving, thr) if len(moving) > thr else moving
        if mode in ("one_agent", "two_agent"):
            picked = self.find_interesting_pair(scn, gt_data_dict, moving)
            return [] if picked is None else (picked[:1] if mode == "one_agent" else picked)
        raise ValueError(f"eval_mode {mode!r}: one_agent, two_agent or multi_agent (cfgs/eval/base.yaml:13-14)")

    def _batched_route_applies(self):
        """The batched route serves this repo's AutoregressivePolicy with the policy's own RTGs (CtRL-Sim, IL, Trajeglish); a foreign
        `Policy`, the Decision Transformer's real-time reward ledger (per-vehicle host bookkeeping) and cfg.eval.batched = False keep
        the per-scenario loop."""
        from ..policies.autoregressive_policy import AutoregressivePolicy
        return (type(self.policy) is AutoregressivePolicy and not self.policy.real_time_rewards and hasattr(self.policy.model, "hip")
                and bool(self.cfg.eval.get("batched", True)))

    def evaluate_policy(self):
        if self._batched_route_applies():
            return self._evaluate_policy_batched()
        return self._evaluate_policy_per_scenario()

    def _evaluate_policy_batched(self):
        """evaluate_policy (policy_evaluator.py:426-576) with the scenario loop turned inside out: the scenes are chosen and their
        vehicles drawn exactly as the per-scenario loop does (same `random` stream), then ALL scenes of equal shape are rolled together
        by one RolloutEngine — one grouping / context / two-pass forward / sampling sequence per step for the whole batch
        (engine.policy_step), the log-replay actions of the vehicles the policy does not control (Evaluator.apply_gt_action,
        evaluators/evaluator.py:160-193: inverse bicycle model against the next logged state) computed for the whole batch in NumPy
        float64 — the same code, hence the same bits, as the per-scenario route — and one simulator step for the batch.  The statistics
        are the per-scenario route's (update_running_statistics on the same arrays): the metric dict is identical to it (1e-12: the
        accumulation order of floating-point sums), at the engine's throughput instead of a host round trip per vehicle and step.

        What this route does NOT do: it never drives the `policy` OBJECT — no `policy.reset()` / `update_state()` / `predict()` calls, so
        the policy's own buffers (`policy.states`, `policy.actions`, `policy.rtgs`, ...) keep whatever an earlier per-scenario session
        left in them; read the rollouts from `last_vehicle_data_dict` / the metric dict, or set `cfg.eval.batched = False` for the
        reference's per-scenario loop.  One RolloutEngine (workspace of `cfg.eval.batch_contexts` contexts) serves all chunks."""
        from .. import discretize as dz
        from ..engine import RolloutEngine
        self.reset()
        pol, w = self.policy, self.cfg_rl_waymo
        T, T1, hsteps = self.steps, self.steps + 1, self.history_steps
        chosen, n_done = [], 0
        for scn, gt_data_dict, moving, pre in self._scenes(self.synthetic):
            if n_done == self.cfg.eval.num_files_to_evaluate // self.cfg.eval.partitions:
                break
            to_eval = self._choose_vehicles(scn, gt_data_dict, moving)
            if not to_eval:
                continue
            n_done += 1
            chosen.append((scn, gt_data_dict, list(to_eval)))
        groups = {}
        for item in chosen:                                    # one engine batch = scenes of equal vehicle and polyline counts
            groups.setdefault((item[0].N, item[0].road_points.shape[0]), []).append(item)
        tilt = (pol.goal_tilt, pol.veh_veh_tilt, pol.veh_edge_tilt) if pol.tilt_dict["tilt"] else (0.0, 0.0, 0.0)
        cap = int(self.cfg.eval.get("batch_scenarios", 256))
        self.batched_scenes = 0
        self._batch_engine = None
        try:
            for (N, _), items in groups.items():
                for c0 in range(0, len(items), cap):
                    self._roll_batch(items[c0:c0 + cap], N, tilt, dz, RolloutEngine, w, T, T1, hsteps)
        finally:
            self._batch_engine = None                          # the workspace goes back to torch's allocator with the evaluation
        return self.compute_metrics()

    def _roll_batch(self, items, N, tilt, dz, RolloutEngine, w, T, T1, hsteps):
        import copy
        pol = self.policy
        S = len(items)
        gt = np.zeros((S, N, T1 + 1, 6))                       # x, y, heading, speed, exist, length (+ one row: apply_gt_action looks at t + 1)
        scns, goal_dicts, evals = [], [], []
        ctrl = np.zeros((S, N), bool)                          # vehicles_to_evaluate
        for k, (scn, gtd, to_eval) in enumerate(items):
            ids = list(range(scn.N))                           # Simulation's vehicle ids are the indices (synthetic) / follow the loader's order
            gd = {}
            for v in ids:
                tr = np.asarray(gtd[v]["traj"], np.float64)
                n = min(len(tr), T1 + 1)
                gt[k, v, :n, :5] = tr[:n, :5]
                gt[k, v, :n, 5] = tr[:n, -1]
                gd[v] = self.initialize_goal_dict(scn, v, tr)
            # the policy sees the goals the evaluator works with (initialize_goal_dict moves the goal of a vehicle that leaves the log)
            s2 = copy.copy(scn)
            s2.goal_pos = np.array([gd[v]["pos"] for v in ids], np.float32)
            s2.goal_heading = np.array([gd[v]["heading"] for v in ids], np.float32)
            s2.goal_speed = np.array([gd[v]["speed"] for v in ids], np.float32)
            # processing order of the vehicles to evaluate: decreasing ground-truth length (autoregressive_policy.py:88-94)
            lengths = [int(np.asarray(gtd[v]["traj"])[:, 4].sum()) for v in to_eval]
            s2.eval_order = np.array(to_eval)[np.argsort(np.array(lengths))[::-1]].astype(np.int32)
            scns.append(s2); goal_dicts.append(gd); evals.append(to_eval)
            ctrl[k, to_eval] = True
        # ONE engine for every group / chunk of the evaluation (its workspace — the expensive allocation — depends on the model batch only;
        # load_scenarios re-sizes the per-scenario tensors for each batch)
        eng = getattr(self, "_batch_engine", None)
        if eng is None:
            eng = self._batch_engine = RolloutEngine(pol.model.cfg, pol.model.weights, pol.model.device,
                                                     max_ctx=int(self.cfg.eval.get("batch_contexts", 256)), seed=int(self.cfg.eval.seed), tilt=tilt,
                                                     temperature=pol.action_temperature, nucleus=pol.nucleus_sampling,
                                                     top_p=pol.nucleus_threshold, model=pol.model.hip, lanes=1)
        eng.load_scenarios(scns, steps=T)
        dev = eng.device
        exist = np.zeros((S, N, T1))
        accel = np.zeros((S, N, T1))
        steer = np.zeros((S, N, T1))
        speeds = np.zeros((S, N, T1), np.float32)
        own_last = np.zeros((T, N), np.int32)                  # last scene of the batch: does a context answer for the vehicle at step t
        import time
        tm = self.batched_timing = {"policy_step_enqueue": 0.0, "read_back": 0.0, "host_actions": 0.0, "upload_and_sim": 0.0}
        for t in range(T):
            t_a = time.perf_counter()
            # update_vehicle_data_dict (:99-159): existence = the log's flag, latched at 0
            exist[:, :, t] = gt[:, :, t, 4] if t == 0 else gt[:, :, t, 4] * (exist[:, :, t - 1] != 0)
            eng.hist_states[:, :, t, 7] = torch.from_numpy(exist[:, :, t].astype(np.float32)).to(dev)
            eng.policy_step(t)
            t_b = time.perf_counter()
            row = eng.hist_states[:, :, t].cpu().numpy()       # synchronises: the step's tokens are sampled
            toks = eng.act_now.cpu().numpy()
            speed = eng.phys[:, :, 16].cpu().numpy()
            bad = eng.nonfinite()
            t_c = time.perf_counter()
            if bad and bad < 65536 and eng.split == "auto" and eng.scheme == 1:
                eng._set_split(0)                              # as predict(): redo the step with the range-safe three-bf16-plane operands
                eng.policy_step(t)
                toks = eng.act_now.cpu().numpy()
                bad = eng.nonfinite()
            if bad:
                raise FloatingPointError(f"{bad} guard events at step {t} of the batched evaluation (csrc/split.h: activation range; csrc/sim.hip: "
                                         f"contact table) in the batch of scenes {[int(getattr(it[0], 'index', -1)) for it in items]}")
            speeds[:, :, t] = speed
            own_last[t] = eng.own_ctx[S - 1].cpu().numpy()
            a = np.zeros((S, N)); st = np.zeros((S, N)); alive = np.ones((S, N), bool)
            by_policy = ctrl & (t >= hsteps - 1)
            # policy.act (autoregressive_policy.py:256-274): a vehicle that does not exist any more is parked; a vehicle no context
            # answers for (dead_agent_veh_ids) gets (0, 0)
            und = dz.undiscretize_actions(np.maximum(toks, 0), w)
            live = by_policy & (exist[:, :, t] != 0)
            a[live], st[live] = np.where(toks[live] >= 0, und[live][:, 0], 0.0), np.where(toks[live] >= 0, und[live][:, 1], 0.0)
            alive[by_policy & (exist[:, :, t] == 0)] = False
            # apply_gt_action (evaluators/evaluator.py:160-193) for everyone else
            rep = ~by_policy
            ok = rep & (gt[:, :, t, 4] != 0) & (gt[:, :, t + 1, 4] != 0) & ~((t > 0) & (exist[:, :, t] == 0))
            if ok.any():
                nxt = np.concatenate([gt[:, :, t + 1, :4][ok], gt[:, :, t + 1, 5][ok][:, None]], 1)
                prev = np.stack([row[..., 0][ok], row[..., 1][ok], row[..., 4][ok], speed[ok]], 1).astype(np.float64)
                a[ok], st[ok] = bicycle_backward(nxt, prev, self.dt)
            alive[rep & ~ok] = False
            accel[:, :, t] = a
            steer[:, :, t] = st
            t_d = time.perf_counter()
            act = torch.from_numpy(np.stack([a, st], -1)).to(dev)
            # what update_state writes back as the action history of step t: the applied action, discretised (identity for a sampled token)
            eng.hist_tok[:, :, t] = torch.from_numpy(dz.discretize_actions(np.stack([a, st], -1), w).astype(np.int32)).to(dev)
            eng.exists.copy_(torch.from_numpy(alive.astype(np.uint8)).to(dev))
            eng.sim_step(t, act)
            t_e = time.perf_counter()
            tm["policy_step_enqueue"] += t_b - t_a; tm["read_back"] += t_c - t_b; tm["host_actions"] += t_d - t_c; tm["upload_and_sim"] += t_e - t_d
        exist[:, :, T] = gt[:, :, T, 4] * (exist[:, :, T - 1] != 0)
        states = eng.hist_states.cpu().numpy()
        coll = eng.coll.cpu().numpy()
        speeds[:, :, T] = eng.phys[:, :, 16].cpu().numpy()
        self.last_vehicle_data_dict = self._vehicle_data_dict_of(S - 1, items[S - 1][0], goal_dicts[S - 1], states, coll, speeds, exist, accel, steer, gt,
                                                                 eng.hist_rtg[S - 1].cpu().numpy(), own_last, dz, w, T)
        for k, (scn, gtd, to_eval) in enumerate(items):
            stt = np.zeros((N, T1, 8))
            stt[..., :5] = states[k, :, :, :5]
            stt[..., 7] = exist[k]
            g = gt[k, :, :T1, :5]
            ids = list(range(N))
            gp = np.array([np.asarray(goal_dicts[k][v]["pos"], np.float64) for v in ids])
            gh = np.array([float(goal_dicts[k][v]["heading"]) for v in ids])
            gs = np.array([float(goal_dicts[k][v]["speed"]) for v in ids])
            self.vehicles_to_evaluate = to_eval
            self.acc.add_scenario(stt, coll[k].astype(np.float64), accel[k], g, gp, gh, gs, self.cfg, eval_ids=[ids.index(v) for v in to_eval])
            self.batched_scenes += 1

    def _vehicle_data_dict_of(self, k, scn, goal_dict, states, coll, speeds, exist, accel, steer, gt, rtg_bins, own, dz, w, T):
        """`last_vehicle_data_dict` of the batched route: scene k's rollout in the per-scenario loop's dict schema
        (policy_evaluator.py:70-96) — what a caller inspecting the evaluator after evaluate_policy() reads."""
        from types import SimpleNamespace as NS
        vdd = {}
        cont = dz.undiscretize_rtgs(rtg_bins, w) if self.policy.predict_rtgs else None
        for v in range(scn.N):
            veh0 = NS(getWidth=lambda v=v: float(scn.width[v]), getLength=lambda v=v: float(scn.length[v]))
            d = self.initialize_vehicle_data_dict(veh0, goal_dict[v])
            norm = np.linalg.norm(np.array([states[k, v, 0, 0], states[k, v, 0, 1]]) - goal_dict[v]["pos"])
            for t in range(T + 1):
                r = states[k, v, t]
                d["position"].append({"x": r[0], "y": r[1]})
                d["velocity"].append({"x": r[2], "y": r[3]})
                d["heading"].append(r[4])
                d["timestep"].append(t)
                d["existence"].append(exist[k, v, t])
                d["gt_position"].append({"x": gt[k, v, t, 0], "y": gt[k, v, t, 1]})
                d["gt_heading"].append(gt[k, v, t, 2])
                d["gt_speed"].append(gt[k, v, t, 3])
                d["acceleration"].append(accel[k, v, t] if t < T else 0)
                d["steering"].append(steer[k, v, t] if t < T else 0)
                veh = NS(position=NS(x=r[0], y=r[1]), speed=speeds[k, v, t], heading=r[4],
                         collision_type_veh=CollisionType.VEHICLE_VEHICLE if coll[k, v, t, 0] else CollisionType.NOT_COLLIDED,
                         collision_type_edge=CollisionType.VEHICLE_ROAD if coll[k, v, t, 1] else CollisionType.NOT_COLLIDED)
                d["reward"].append(self.compute_reward(veh, goal_dict[v], norm, d))
                if cont is not None and t < T:
                    d[self.policy.key_dict["rtgs"]].append(np.array(cont[v, t]) if own[t, v] >= 0 else
                                                           np.array([0] * self.policy.cfg_model.num_reward_components))
            vdd[v] = d
        return vdd

    def _evaluate_policy_per_scenario(self):
        self.reset()
        n_done = 0
        for scn, gt_data_dict, moving, pre in self._scenes(self.synthetic):
            if n_done == self.cfg.eval.num_files_to_evaluate // self.cfg.eval.partitions:
                break
            self.policy.scenario_index = scn.index
            sim = Simulation(scn, device=self.policy.model.device, steps=self.steps, dt=self.dt)
            vehicles = sim.getScenario().vehicles()
            for veh in vehicles:
                veh.expert_control = False
                veh.physics_simulated = True
            self.vehicles_to_evaluate = self._choose_vehicles(scn, gt_data_dict, moving)
            if not self.vehicles_to_evaluate:
                continue
            n_done += 1
            self.road_edge_polylines = self.extract_road_edge_polylines(scn)
            preproc_data = {"road_points": scn.road_points.astype(np.float64), "road_types": scn.road_types.copy()}
            if pre is not None:
                preproc_data["rtgs"] = pre["rtgs"]
            vdd, goal_dict, goal_norm = {}, {}, {}
            for veh in vehicles:
                v = veh.getID()
                goal_dict[v] = self.initialize_goal_dict(scn, v, np.array(gt_data_dict[v]["traj"]))
                vdd[v] = self.initialize_vehicle_data_dict(veh, goal_dict[v])
                goal_norm[v] = np.linalg.norm(np.array([veh.getPosition().x, veh.getPosition().y]) - goal_dict[v]["pos"])
            self.policy.reset(vdd)
            for t in range(self.steps):
                vdd = self.update_vehicle_data_dict(t, vehicles, vdd, goal_dict, goal_norm, gt_data_dict, preproc_data)
                self.policy.update_state(vdd, self.vehicles_to_evaluate, t)
                vdd = self.policy.predict(vdd, gt_data_dict, preproc_data, None, self.vehicles_to_evaluate, t)
                for veh in vehicles:
                    v = veh.getID()
                    if t >= self.history_steps - 1 and v in self.vehicles_to_evaluate:
                        veh, act = self.policy.act(veh, t, vdd)
                    else:
                        veh, act = self.apply_gt_action(veh, t, gt_data_dict, vdd)
                    vdd[v]["acceleration"].append(act[0])
                    vdd[v]["steering"].append(act[1])
                sim.step(self.dt)
            vdd = self.update_vehicle_data_dict(self.steps, vehicles, vdd, goal_dict, goal_norm, gt_data_dict, preproc_data)
            for veh in vehicles:
                vdd[veh.getID()]["acceleration"].append(0)
                vdd[veh.getID()]["steering"].append(0)
            self.last_vehicle_data_dict = vdd
            self.update_running_statistics(vdd, scn, gt_data_dict, goal_dict)
        return self.compute_metrics()

    # ---- policy_evaluator.py:162-248 on arrays
    def update_running_statistics(self, vdd, scn, gt_data_dict, goal_dict=None):
        ids = list(vdd.keys())
        T1 = self.steps + 1
        st = np.zeros((len(ids), T1, 8))
        coll = np.zeros((len(ids), T1, 2))
        accel = np.zeros((len(ids), T1))
        gt = np.zeros((len(ids), T1, 5))
        for i, v in enumerate(ids):
            d = vdd[v]
            st[i, :, 0] = [p["x"] for p in d["position"]]; st[i, :, 1] = [p["y"] for p in d["position"]]
            st[i, :, 2] = [p["x"] for p in d["velocity"]]; st[i, :, 3] = [p["y"] for p in d["velocity"]]
            st[i, :, 4] = d["heading"]; st[i, :, 7] = d["existence"]
            r = np.array(d["reward"])
            coll[i] = r[:, 6:8]
            accel[i] = d["acceleration"]
            tr = gt_data_dict[v]["traj"]
            gt[i] = tr[:, [0, 1, 2, 3, 4]]
        # goals as the loop used them (initialize_goal_dict moves the goal of a vehicle that leaves the log to its last logged pose;
        # the reference reads the 'goal' flag from the rewards computed with that goal_dict, policy_evaluator.py:162-186)
        if goal_dict is None:
            goal_dict = {v: self.initialize_goal_dict(scn, v, np.array(gt_data_dict[v]["traj"])) for v in ids}
        gp = np.array([np.asarray(goal_dict[v]["pos"], np.float64) for v in ids])
        gh = np.array([float(goal_dict[v]["heading"]) for v in ids])
        gs = np.array([float(goal_dict[v]["speed"]) for v in ids])
        self.acc.add_scenario(st, coll, accel, gt, gp, gh, gs, self.cfg, eval_ids=[ids.index(v) for v in self.vehicles_to_evaluate])

    def compute_metrics(self):
        return self.acc.compute()
