"""`AutoregressivePolicy` — drop-in for the reference's policies/autoregressive_policy.py:9-274, backed by the HIP path.

Same constructor, `predict(vehicle_data_dict, gt_data_dict, preproc_data, dset, vehicles_to_evaluate, t)` and
`act(veh, t, vehicle_data_dict)` contracts: predict sets `next_acceleration` / `next_steering` / `next_rtg_*` and
appends to the `rtgs` list of every vehicle; act drives a vehicle handle exposing getID / setPosition / .acceleration=
/ brake() / .steering= (the pybind Vehicle surface, nocturne/pybind11/src/object.cc:33-99, vehicle.cc:19-21).

What happens inside predict is not the reference's NumPy + per-group model calls: the host buffers of `Policy` are
mirrored into a one-scenario device session and the whole of get_data / two-pass model / sampling runs as the kernel
sequence of `ctrlsim_amd.engine.RolloutEngine.policy_step` (focal grouping, context build, forward pass 1, RTG race,
forward pass 2, action race).  Sampling noise is the counter-based Exp(1) stream keyed by
(cfg.eval.seed, scenario index, t, vehicle index, head) instead of torch's global generator.
"""
from __future__ import annotations

import numpy as np

from .policy import Policy
from .. import discretize as dz
from ..scenarios import Scenario


class AutoregressivePolicy(Policy):
    def __init__(self, cfg, model_path, model, use_rtg, predict_rtgs, discretize_rtgs, real_time_rewards,
                 privileged_return, max_return, min_return, key_dict, tilt_dict, name, action_temperature,
                 nucleus_sampling, nucleus_threshold):
        super().__init__(cfg, model_path, model, use_rtg, predict_rtgs, discretize_rtgs, real_time_rewards,
                         privileged_return, max_return, min_return, key_dict, tilt_dict, name)
        self.action_temperature = action_temperature
        self.nucleus_sampling = nucleus_sampling
        self.nucleus_threshold = nucleus_threshold
        if tilt_dict["tilt"]:
            self.goal_tilt = tilt_dict["goal_tilt"]
            self.veh_veh_tilt = tilt_dict["veh_veh_tilt"]
            self.veh_edge_tilt = tilt_dict["veh_edge_tilt"]
        variant = model.dims.VARIANT
        if variant == 0 and (not (use_rtg and predict_rtgs and discretize_rtgs) or real_time_rewards):
            raise NotImplementedError("the CtRL-Sim model runs as in cfgs/policy/ctrl_sim.yaml: use_rtg, predict_rtgs, "
                                      "discretize_rtgs, no real_time_rewards")
        if variant in (1, 2) and (use_rtg or predict_rtgs or real_time_rewards):
            raise NotImplementedError("the IL / Trajeglish models have no RTG tokens (cfgs/policy/{il,trajeglish}.yaml: "
                                      "use_rtg = predict_rtgs = False)")
        if variant == 3 and not (use_rtg and real_time_rewards and not predict_rtgs and not discretize_rtgs):
            raise NotImplementedError("the Decision-Transformer model runs as in cfgs/policy/dt.yaml: use_rtg, real_time_rewards, "
                                      "continuous (not discretised, not predicted) RTGs")
        self._session = None
        self._synced_t = None
        self.scenario_index = 0

    # ------------------------------------------------------------------ device session
    def _open_session(self, vehicle_data_dict, preproc_data, gt_data_dict, vehicles_to_evaluate):
        from ..engine import RolloutEngine
        ids = list(vehicle_data_dict.keys())
        n = len(ids)
        f32 = lambda a: np.asarray(a, np.float32)
        d0 = [vehicle_data_dict[v] for v in ids]
        # processing order of the vehicles to evaluate: decreasing ground-truth length (autoregressive_policy.py:88-94)
        lengths = [int(np.array(gt_data_dict[v]["traj"])[:, 4].sum()) for v in vehicles_to_evaluate]
        order = list(np.array(vehicles_to_evaluate)[np.argsort(np.array(lengths))[::-1]])
        eval_order = np.array([self.veh_id_to_idx[v] for v in order], np.int32)
        rp = np.asarray(preproc_data["road_points"])
        scn = Scenario(index=self.scenario_index, length=f32([d["length"] for d in d0]), width=f32([d["width"] for d in d0]),
                       x=f32([d["position"][0]["x"] for d in d0]), y=f32([d["position"][0]["y"] for d in d0]),
                       heading=f32([d["heading"][0] for d in d0]), speed=f32(np.zeros(n)),
                       goal_pos=f32([[d["goal_position"]["x"], d["goal_position"]["y"]] for d in d0]),
                       goal_heading=f32([d["goal_heading"] for d in d0]), goal_speed=f32([d["goal_speed"] for d in d0]),
                       types=self.types.copy(), road_points=f32(rp), road_types=np.asarray(preproc_data["road_types"], np.float64),
                       edge_segments=np.zeros((0, 4), np.float32), eval_order=eval_order)
        tilt = (self.goal_tilt, self.veh_veh_tilt, self.veh_edge_tilt) if self.tilt_dict["tilt"] else (0.0, 0.0, 0.0)
        eng = RolloutEngine(self.model.cfg, self.model.weights, self.model.device, max_ctx=max(16, n),
                            seed=int(self.cfg.eval.seed), tilt=tilt, temperature=self.action_temperature,
                            nucleus=self.nucleus_sampling, top_p=self.nucleus_threshold, model=self.model.hip)
        eng.device_ledger = False      # real_time_rewards: the evaluator's own bookkeeping feeds hist_rtg (predict below)
        eng.load_scenarios([scn], steps=self.steps)
        self._session = eng
        self._session_key = (tuple(ids), rp.shape)

    def reset(self, vehicle_data_dict):
        super().reset(vehicle_data_dict)
        self._session = None
        self._synced_t = None

    def _sync_session(self, vehicle_data_dict, gt_data_dict, preproc_data, vehicles_to_evaluate, t):
        """Open the device session at t == 0 (or when none is open) and mirror the host history buffers into it.  A session that was
        synchronised at step t - 1 receives only the rows `Policy.update_state` wrote since — the state row of step t, the action / RTG
        rows of step t - 1 (and the RTG row of step t under real_time_rewards), policies/policy.py:68-105 — instead of the whole
        [N, steps] history at every step (round-4 review: the full re-upload was most of the plugin route's per-step cost); any other
        call order falls back to the full mirror."""
        import torch
        w = self.cfg_rl_waymo
        fresh = self._session is None or t == 0
        if fresh:
            if vehicle_data_dict is None:        # get_data without a preceding predict: what reset() / update_state() hold
                vehicle_data_dict = self._dict_from_buffers()
            self._open_session(vehicle_data_dict, preproc_data, gt_data_dict, vehicles_to_evaluate)
            self._synced_t = None
        eng = self._session
        dev = eng.device
        n = self.states.shape[0]
        dt3 = self.model.dims.VARIANT == 3

        def rtg_rows(rows):                                  # continuous RTGs, clip-normalised (get_data:73-78), as float bits (DT) / bins
            if dt3:
                from ..rewards import normalize_rtgs
                return np.ascontiguousarray(normalize_rtgs(rows, w), np.float32).view(np.int32)
            return dz.discretize_rtgs_from_raw(rows, w).astype(np.int32)

        if self._synced_t is not None and t == self._synced_t + 1 and 0 < t < self.steps:
            eng.hist_states[0, :, t].copy_(torch.from_numpy(self.states[:, t].astype(np.float32)).to(dev))
            eng.hist_tok[0, :, t - 1].copy_(torch.from_numpy(dz.discretize_actions(self.actions[:, t - 1], w).astype(np.int32)).to(dev))
            lo = t - 1
            hi = t + 1 if (self.real_time_rewards and self.use_rtg) else t
            eng.hist_rtg[0, :, lo:hi].copy_(torch.from_numpy(np.ascontiguousarray(rtg_rows(self.rtgs[:, lo:hi]))).to(dev))
        else:
            hs = np.zeros((1, n, self.steps + 1, 8), np.float32)
            hs[0, :, :self.steps] = self.states
            eng.hist_states.copy_(torch.from_numpy(hs).to(dev))
            eng.hist_tok.copy_(torch.from_numpy(dz.discretize_actions(self.actions, w).astype(np.int32)[None]).to(dev))
            eng.hist_rtg.copy_(torch.from_numpy(np.ascontiguousarray(rtg_rows(self.rtgs))[None]).to(dev))
            eng.goals.copy_(torch.from_numpy(self.goals[:, 0][None]).to(dev))
        self._synced_t = t
        return eng

    def _dict_from_buffers(self):
        """The per-vehicle fields _open_session reads, rebuilt from the Policy buffers (row 0)."""
        out = {}
        kinds = ("unset", "vehicle", "pedestrian", "cyclist", "other")
        for i, v in self.idx_to_veh_id.items():
            g = self.goals[i, 0]
            out[v] = {"length": self.states[i, 0, 5], "width": self.states[i, 0, 6], "position": [{"x": self.states[i, 0, 0], "y": self.states[i, 0, 1]}],
                      "heading": [self.states[i, 0, 4]], "goal_position": {"x": g[0], "y": g[1]}, "goal_heading": g[4] if len(g) > 4 else 0.0,
                      "goal_speed": float(np.hypot(g[2], g[3])) if len(g) > 3 else 0.0, "type": kinds[int(np.argmax(self.types[i]))]}
        return out

    def get_data(self, gt_data_dict, preproc_data, dset, vehicles_to_evaluate, t, vehicle_data_dict=None):
        """The reference's return contract (autoregressive_policy.py:51-165): (motion_datas {focal veh id: {'agent': {agent_states
        [1,A,T,8], agent_types [1,A,5], goals [1,A,5], actions [1,A,T] token ids, rtgs [1,A,T,3] bins, timesteps [1,A,T],
        moving_agent_mask [1,A]}, 'map': {road_points [1,P,NP,3], road_types [1,P,8]}} as torch tensors}, dead_agent_veh_ids,
        new_agent_idx_dicts {focal: {agent index: context slot}}, data_veh_ids {focal: [veh ids it answers for]}).  The tensors
        are built by the device kernels predict() runs (ctrlsim_group_build / ctrlsim_build_context, the plain 24-slot layout) and
        copied back; `dset` is not used.  The context membership the policy persists from step to step is left as it was, so a
        call before (or instead of) predict() does not change what predict() does."""
        import ctypes as C
        import torch
        from .. import _lib
        from ..engine import CtxBuffers
        eng = self._sync_session(vehicle_data_dict, gt_data_dict, preproc_data, vehicles_to_evaluate, t)
        lib, p, st, d = eng.lib, _lib.ptr, _lib.stream_ptr(), eng.dims
        N, Tmax, w = eng.N, eng.steps, self.cfg_rl_waymo
        persist0 = eng.persist.clone()
        _lib.check(lib.ctrlsim_group_build(1, N, d.A, d.T, t, Tmax + 1, float(w.agent_dist_threshold), p(eng.hist_states),
                                           p(eng.eval_order), 1 if eng.P_all > 0 else 0, p(eng.persist), p(eng.n_groups),
                                           p(eng.grp_focal), p(eng.grp_ids), p(eng.grp_members), p(eng.own_g), p(eng.mem_g),
                                           p(eng.tilted), st), "group_build")
        G = int(eng.n_groups.cpu()[0])
        _lib.check(lib.ctrlsim_ctx_index(0, 1, N, p(eng.n_groups), p(eng.grp_focal), p(eng.grp_ids), p(eng.own_g), p(eng.mem_g),
                                         p(eng.ctx_scn), p(eng.ctx_grp), p(eng.own_ctx), p(eng.own_slot), p(eng.mem_ctx),
                                         p(eng.mem_slot), p(eng.ctx_base), st), "ctx_index")
        cb = CtxBuffers(d, max(G, 1), eng.device)
        zero4 = (C.c_int * 4)(*eng._zero4)
        if G:
            _lib.check(lib.ctrlsim_build_context(G, N, d.A, d.T, t, d.T, 0, Tmax + 1, Tmax, eng.P_all, d.P, d.NP, p(eng.ctx_scn),
                                                 p(eng.ctx_grp), p(eng.grp_focal), p(eng.grp_ids), p(eng.hist_states),
                                                 p(eng.hist_tok), p(eng.hist_rtg), p(eng.goals), p(eng.types), p(eng.roads),
                                                 p(eng.rtypes), zero4, C.byref(cb.struct), st), "build_context")
        torch.cuda.synchronize(eng.device)
        focal = eng.grp_focal.cpu().numpy()[0, :G]
        ids = eng.grp_ids.cpu().numpy().astype(np.uint64)[0, :G]
        mem = eng.grp_members.cpu().numpy().astype(np.uint64)[0, :G]
        mem_g = eng.mem_g.cpu().numpy()[0]
        eng.persist.copy_(persist0)
        bits = lambda m: [i for i in range(64) if (int(m) >> i) & 1]
        order = {int(v): k for k, v in enumerate(eng.eval_order.cpu().numpy()[0]) if v >= 0}
        moving = np.linalg.norm(self.states[:, 0, :2] - self.goals[:, 0, :2], axis=1) > w.moving_threshold
        st12 = cb.st12.cpu().numpy().reshape(-1)[:G * d.T * d.A * 12].reshape(G, d.T, d.A, 12)
        ex = cb.exist.cpu().numpy().reshape(-1)[:G * d.T * d.A].reshape(G, d.T, d.A)
        tok = cb.act_tok.cpu().numpy().reshape(-1)[:G * d.T * d.A].reshape(G, d.T, d.A)
        rb = cb.rtg_bin.cpu().numpy().reshape(-1)[:G * d.T * d.A * 3].reshape(G, d.T, d.A, 3)
        ts = cb.tstep.cpu().numpy().reshape(-1)[:G * d.T].reshape(G, d.T)
        g5, rp, rt = cb.goal5.cpu().numpy()[:G], cb.road_pts.cpu().numpy()[:G], cb.road_types.cpu().numpy()[:G]
        motion_datas, idx_dicts, data_veh_ids = {}, {}, {}
        for gi in range(G):
            fid = self.idx_to_veh_id[int(focal[gi])]
            slots = bits(ids[gi])                                       # agent indices in slot order
            states = np.concatenate([st12[gi, :, :, :7], ex[gi][..., None]], -1).transpose(1, 0, 2)       # [A,T,8]
            mm = np.zeros(d.A, bool)
            mm[:len(slots)] = moving[slots]
            rtgs = rb[gi].transpose(1, 0, 2)
            if self.model.dims.VARIANT == 3:
                rtgs = np.ascontiguousarray(rtgs).view(np.float32)
            tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)[None])
            motion_datas[fid] = {"agent": {"agent_states": tt(states), "agent_types": tt(st12[gi, 0, :, 7:]), "goals": tt(g5[gi]),
                                           "actions": tt(tok[gi].T), "rtgs": tt(rtgs),
                                           "timesteps": tt(np.repeat(ts[gi][None], d.A, 0)), "moving_agent_mask": tt(mm)},
                                 "map": {"road_points": tt(rp[gi]), "road_types": tt(rt[gi])}}
            idx_dicts[fid] = {a: k for k, a in enumerate(slots)}
            others = sorted((a for a in bits(mem[gi]) if a != int(focal[gi])), key=lambda a: order.get(a, 1 << 30))
            data_veh_ids[fid] = [fid] + [self.idx_to_veh_id[a] for a in others]
        dead = [v for v in vehicles_to_evaluate if mem_g[self.veh_id_to_idx[v]] < 0]
        return motion_datas, dead, idx_dicts, data_veh_ids

    def predict(self, vehicle_data_dict, gt_data_dict, preproc_data, dset, vehicles_to_evaluate, t):
        import torch
        w = self.cfg_rl_waymo
        eng = self._sync_session(vehicle_data_dict, gt_data_dict, preproc_data, vehicles_to_evaluate, t)
        dev = eng.device
        n = self.states.shape[0]
        eng.policy_step(t)
        bad = eng.nonfinite()                              # NaN logits (fp16 overflow of the split operands, bad weights) must not
        if bad >= 65536:                                   # simulator events: the second word of the guard pair, reported in the high half (csrc/common.h): not a matter of the split
            raise FloatingPointError(f"{bad >> 16} simulator contacts beyond the island solver's table before step {t} (csrc/sim.hip)")
        if bad and eng.split == "auto" and eng.scheme == 1:
            eng._set_split(0)                              # pass as "rtg bin 0 / zero action": redo the step with the range-safe
            eng.policy_step(t)                             # three-bf16-plane operands (they stay selected for this model: every
            bad = eng.nonfinite()                          # later session starts on them), csrc/split.h
        if bad:
            raise FloatingPointError(f"{bad} sampling races had no finite logit at step {t} (csrc/split.h: activation range)")
        bins = eng.hist_rtg[0, :, t].cpu().numpy()
        toks = eng.act_now[0].cpu().numpy()
        own = eng.own_ctx[0].cpu().numpy()
        cont_rtg = dz.undiscretize_rtgs(bins, w)
        ids = list(vehicle_data_dict.keys())
        for i, v in enumerate(ids if self.predict_rtgs else []):       # autoregressive_policy.py:242-247
            d = vehicle_data_dict[v]
            if own[i] >= 0:
                d["next_rtg_goal"], d["next_rtg_veh"], d["next_rtg_road"] = cont_rtg[i]
                d[self.key_dict["rtgs"]].append(np.array(cont_rtg[i]))
            else:
                d[self.key_dict["rtgs"]].append(np.array([0] * self.cfg_model.num_reward_components))
        for v in vehicles_to_evaluate:
            i = self.veh_id_to_idx[v]
            if toks[i] >= 0:
                a, s = dz.undiscretize_actions(np.array([toks[i]]), w)[0]
            else:                                                    # dead_agent_veh_ids
                a, s = 0.0, 0.0
            vehicle_data_dict[v][self.key_dict["next_acceleration"]] = a
            vehicle_data_dict[v][self.key_dict["next_steering"]] = s
        return vehicle_data_dict

    def act(self, veh, t, vehicle_data_dict):
        veh_id = veh.getID()
        veh_exists = vehicle_data_dict[veh_id]["existence"][-1]
        if not veh_exists:
            acceleration, steering = 0.0, 0.0
            veh.setPosition(-1000000, -1000000)
        else:
            acceleration = vehicle_data_dict[veh_id][self.key_dict["next_acceleration"]]
            steering = vehicle_data_dict[veh_id][self.key_dict["next_steering"]]
        if acceleration > 0.0:
            veh.acceleration = acceleration
        else:
            veh.brake(np.abs(acceleration))
        veh.steering = steering
        return veh, [acceleration, steering]
