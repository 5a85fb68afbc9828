"""Batched closed-loop rollout engine: S independent scenarios resident in HBM, stepped by HIP kernels.

This is the MI355X-native replacement of the reference's per-scenario Python loop
(evaluators/policy_evaluator.py:514-557: update dict -> Policy.update_state -> AutoregressivePolicy.predict ->
act -> Simulation.step).  All per-step state lives on the device ([S, N, ...] arrays, see include/ctrlsim.h); the host
only sequences kernel launches through the C ABI and reads ONE small array per step (groups per scenario, needed to
size the model batch).  torch is used for device memory and streams only.

Per step t:
  1. ctrlsim_group_build       focal groups of every scenario (one wavefront per scenario)
  2. for chunks of scenarios whose groups fit the model batch:
       ctrlsim_ctx_index / ctrlsim_build_context     agent-local context tensors (float64 SE(2), nearest-P polylines)
       ctrlsim_dt_forward_pass1  -> RTG logits       (map encoder, scene encoder, decoder; K/V cached)
       ctrlsim_sample_rtg                            first-owner rule, tilt, exponential race
       ctrlsim_dt_forward_pass2  -> action logits    (only the A RTG tokens are re-evaluated)
       ctrlsim_sample_action
  3. ctrlsim_sim_step          FreeCar/Box2D-equivalent update, collision flags, history append
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, pack as _pack
from .spec import Dims, ZERO_ACTION_TOKEN, ZERO_RTG_BINS


def _dims_struct(d: Dims):
    return _lib.Dims(A=d.A, T=d.T, P=d.P, NP=d.NP, D=d.D, H=d.H, F=d.F, V=d.V, R=d.R, C=d.C, NE=d.NE, ND=d.ND,
                     MAXT=d.MAXT, variant=d.VARIANT)


class HipModel:
    """Device-resident weights + the opaque ctrlsim_model handle."""

    def __init__(self, cfg, weights: dict, device="cuda:0"):
        self.cfg = cfg
        self.dims = Dims(cfg)
        self.device = torch.device(device)
        self.lib = _lib.lib()
        flat, names, offsets = _pack.pack(self.dims, weights)
        self.flat = torch.from_numpy(flat).to(self.device)
        self._names = (C.c_char_p * len(names))(*[n.encode() for n in names])
        self._offsets = (C.c_int64 * len(names))(*[int(o) for o in offsets])
        self.cdims = _dims_struct(self.dims)
        h = C.c_void_p()
        _lib.check(self.lib.ctrlsim_model_create(C.byref(self.cdims), self.flat.data_ptr(), len(names), self._names,
                                                 self._offsets, C.byref(h)), "model_create")
        self.handle = h

    def workspace_bytes(self, B, Tq):
        n = self.lib.ctrlsim_forward_workspace_bytes(C.byref(self.cdims), B, Tq)
        if n < 0:
            raise RuntimeError(f"workspace query failed: {n}")
        return int(n)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ctrlsim_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class CtxBuffers:
    """Context tensors of up to Bmax contexts (the ctrlsim_ctx struct)."""

    def __init__(self, d: Dims, Bmax, device):
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)
        self.st12 = z(Bmax, d.T, d.A, 12)
        self.exist = z(Bmax, d.T, d.A)
        self.goal5 = z(Bmax, d.A, 5)
        self.act_tok = z(Bmax, d.T, d.A, dt=torch.int32)
        self.rtg_bin = z(Bmax, d.T, d.A, 3, dt=torch.int32)
        self.tstep = z(Bmax, d.T, dt=torch.int32)
        self.slot_gid = z(Bmax, d.A, dt=torch.int32)
        self.road_pts = z(Bmax, d.P, d.NP, 3)
        self.road_types = z(Bmax, d.P, 8)
        self.struct = _lib.Ctx(*(getattr(self, k).data_ptr() for k in ("st12", "exist", "goal5", "act_tok", "rtg_bin",
                                                                       "tstep", "slot_gid", "road_pts", "road_types")))


def ctx_from_reference_layout(d: Dims, data: dict, Tq: int, device):
    """Reference-layout model inputs (agent_states [B,A,T,8], ... as in modules/encoder.py:52-63) -> CtxBuffers with
    the first Tq window steps.  Used by the model-level parity tests."""
    B = data["agent_states"].shape[0]
    cb = CtxBuffers(d, B, device)
    st = np.asarray(data["agent_states"], np.float64)
    types = np.broadcast_to(np.asarray(data["agent_types"], np.float64)[:, :, None, :], (B, d.A, d.T, 5))
    st12 = np.concatenate([st[..., :7], types], -1).transpose(0, 2, 1, 3)[:, :Tq]
    flat = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(device).reshape(-1)
    cb.st12.view(-1)[:B * Tq * d.A * 12] = flat(st12, np.float32)
    cb.exist.view(-1)[:B * Tq * d.A] = flat(st[..., 7].transpose(0, 2, 1)[:, :Tq], np.float32)
    cb.goal5.copy_(torch.from_numpy(np.asarray(data["goals"], np.float64).astype(np.float32)).to(device))
    cb.act_tok.view(-1)[:B * Tq * d.A] = flat(np.asarray(data["actions"]).transpose(0, 2, 1)[:, :Tq], np.int32)
    rt = np.asarray(data["rtgs"]).transpose(0, 2, 1, 3)[:, :Tq]
    if d.VARIANT == 3:                                   # decision transformer: continuous RTGs travel as float bits
        rt = np.ascontiguousarray(rt, np.float32).view(np.int32)
    cb.rtg_bin.view(-1)[:B * Tq * d.A * 3] = flat(rt, np.int32)
    cb.tstep.view(-1)[:B * Tq] = flat(np.asarray(data["timesteps"])[:, 0, :Tq, 0], np.int32)
    cb.slot_gid.fill_(-1)
    cb.road_pts.copy_(torch.from_numpy(np.asarray(data["road_points"], np.float64).astype(np.float32)).to(device))
    cb.road_types.copy_(torch.from_numpy(np.asarray(data["road_types"], np.float64).astype(np.float32)).to(device))
    return cb


class RolloutEngine:
    def __init__(self, cfg, weights: dict, device="cuda:0", max_ctx=256, seed=0, tilt=(0.0, 0.0, 0.0),
                 temperature=None, nucleus=None, top_p=None, kinematic=False, model=None, use_cache=True, contacts=True):
        self.cfg = cfg
        self.w = cfg.dataset.waymo
        self.dims = Dims(cfg)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.model = model if model is not None else HipModel(cfg, weights, device)
        self.lib = self.model.lib
        pol = cfg.eval.policy
        self.temperature = float(pol.action_temperature if temperature is None else temperature)
        nuc = bool(pol.nucleus_sampling if nucleus is None else nucleus)
        self.top_p = float(pol.nucleus_threshold if top_p is None else top_p) if nuc else 0.0
        self.seed = int(seed)
        # tilt = (goal, veh_veh, veh_edge) for every scenario, or an [S,3] array: one triple per scenario (tilt sweep)
        tilt = np.asarray(tilt, np.float64)
        self._tilt_per_scenario = tilt if tilt.ndim == 2 else None
        self.tilt_scn = None
        self.tilt = (C.c_double * 3)(*([0.0, 0.0, 0.0] if tilt.ndim == 2 else [float(x) for x in tilt]))
        self.kinematic = int(bool(kinematic))
        self.max_ctx = int(max_ctx)
        self.use_cache = bool(use_cache) and not self.dims.VARIANT    # the K/V-cached phase is built for the CtRL-Sim tokens
        self.contacts = bool(contacts) and not kinematic
        self.dt = float(cfg.nocturne.dt)
        w = self.w
        self.disc6 = (C.c_double * 6)(w.min_accel, w.max_accel, w.min_steer, w.max_steer, w.accel_discretization,
                                      w.steer_discretization)
        # "not yet written" rows: zero action, zero RTG -> bins (0, 35, 35); the Decision-Transformer variant carries continuous,
        # normalised RTGs as float bits: zero -> (0, 0.1, 0.1) (autoregressive_policy.py:73-78)
        self.zero_rtg = ZERO_RTG_BINS
        if self.dims.VARIANT == 3:
            from .rewards import normalize_rtgs
            self.zero_rtg = tuple(int(v) for v in normalize_rtgs(np.zeros(3), self.w).astype(np.float32).view(np.int32))
        self._zero4 = (C.c_int * 4)(ZERO_ACTION_TOKEN, *self.zero_rtg)
        self.ctx = CtxBuffers(self.dims, self.max_ctx, self.device)
        self.ws = torch.empty(self.model.workspace_bytes(self.max_ctx, self.dims.T), dtype=torch.uint8, device=self.device)
        d = self.dims
        self.rtg_logits = torch.empty(self.max_ctx, d.A, d.R * d.C, device=self.device)
        self.act_logits = torch.empty(self.max_ctx, d.A, d.V, device=self.device)
        self.S = 0

    # ------------------------------------------------------------------ scenario upload / reset
    def load_scenarios(self, scns, steps=None):
        dev, d = self.device, self.dims
        S = len(scns)
        N = scns[0].N
        assert all(s.N == N for s in scns) and N <= 64, "one batch = equal N <= 64 vehicles per scenario"
        P_all = scns[0].road_points.shape[0]
        assert all(s.road_points.shape[0] == P_all for s in scns)
        self.S, self.N, self.P_all = S, N, P_all
        self.steps = int(steps if steps is not None else self.cfg.nocturne.steps)
        Tmax, Tmax1 = self.steps, self.steps + 1
        E = max(1, max(len(s.edge_segments) for s in scns))
        edges = np.full((S, E, 4), 1e30, np.float32)
        for i, s in enumerate(scns):
            edges[i, :len(s.edge_segments)] = s.edge_segments
        self.E = E
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        self.init_pose = f32(np.stack([np.stack([s.x, s.y, s.heading, s.speed], 1) for s in scns]))
        self.size = f32(np.stack([np.stack([s.length, s.width], 1) for s in scns]))
        self.edges = f32(edges)
        self.exists = torch.ones(S, N, dtype=torch.uint8, device=dev)
        self.goals = torch.from_numpy(np.stack([s.goals5() for s in scns])).to(dev)           # float64
        self.types = f32(np.stack([s.types for s in scns]))
        self.roads = f32(np.stack([s.road_points for s in scns]))
        self.rtypes = f32(np.stack([s.road_types for s in scns]))
        eo = -np.ones((S, N), np.int32)
        for i, s in enumerate(scns):
            eo[i, :len(s.eval_order)] = s.eval_order
        self.eval_order = torch.from_numpy(eo).to(dev)
        self.scenario_id = torch.tensor([s.index for s in scns], dtype=torch.int64, device=dev)
        if self._tilt_per_scenario is not None:
            assert self._tilt_per_scenario.shape == (S, 3), "per-scenario tilt must be [S,3]"
            self.tilt_scn = torch.from_numpy(np.ascontiguousarray(self._tilt_per_scenario)).to(dev)
        z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device=dev)
        self.phys = z(S, N, 20)
        # Box2D contact manifolds + impulses per vehicle pair (None = contact-free integration)
        self.contact_state = z(S, int(self.lib.ctrlsim_sim_contact_floats(N))) if self.contacts else None
        self.hist_states = z(S, N, Tmax1, 8)
        self.coll = z(S, N, Tmax1, 2, dt=torch.uint8)
        self.hist_tok = z(S, N, Tmax, dt=torch.int32)
        self.hist_rtg = z(S, N, Tmax, 3, dt=torch.int32)
        self.act_now = z(S, N, dt=torch.int32)
        self.applied = z(S, N, Tmax, 2, dt=torch.float64)
        self.persist = z(S, N, dt=torch.int64)
        self.n_groups = z(S, dt=torch.int32)
        self.n_groups_host = torch.zeros(S, dtype=torch.int32).pin_memory()
        self.grp_focal = z(S, N, dt=torch.int32)
        self.grp_ids = z(S, N, dt=torch.int64)
        self.grp_members = z(S, N, dt=torch.int64)
        self.own_g, self.mem_g = z(S, N, dt=torch.int32), z(S, N, dt=torch.int32)
        self.tilted = z(S, N, dt=torch.uint8)
        self.own_ctx, self.own_slot = z(S, N, dt=torch.int32), z(S, N, dt=torch.int32)
        self.mem_ctx, self.mem_slot = z(S, N, dt=torch.int32), z(S, N, dt=torch.int32)
        self.ctx_base = z(S, dt=torch.int32)
        self.ctx_scn, self.ctx_grp = z(self.max_ctx, dt=torch.int32), z(self.max_ctx, dt=torch.int32)
        self.groups_per_step = np.zeros((self.steps, S), np.int32)
        self.reset()

    def reset(self):
        st = _lib.stream_ptr()
        self.lib.ctrlsim_nonfinite_count(1)
        self.hist_states.zero_()
        self.coll.zero_()
        self.hist_tok.fill_(ZERO_ACTION_TOKEN)
        self.hist_rtg.copy_(torch.tensor(self.zero_rtg, dtype=torch.int32, device=self.device).expand_as(self.hist_rtg))
        self.persist.zero_()
        self.applied.zero_()
        p = _lib.ptr
        _lib.check(self.lib.ctrlsim_sim_init(self.S, self.N, self.E, p(self.init_pose), p(self.size), p(self.edges),
                                             p(self.exists), p(self.phys), p(self.hist_states), p(self.coll),
                                             self.steps + 1, p(self.contact_state), st), "sim_init")

    # ------------------------------------------------------------------ one step
    def _chunks(self, counts):
        chunks, s0, acc = [], 0, 0
        for s, c in enumerate(counts):
            c = int(c)
            if c > self.max_ctx:
                raise RuntimeError(f"scenario {s} has {c} focal groups > max_ctx={self.max_ctx}")
            if acc + c > self.max_ctx or s - s0 >= 4095:
                chunks.append((s0, s, acc))
                s0, acc = s, 0
            acc += c
        chunks.append((s0, len(counts), acc))
        return [c for c in chunks if c[1] > c[0]]

    def step(self, t, noise_rtg=None, noise_act=None):
        """One closed-loop step: policy (grouping, contexts, two-pass model, sampling) then the simulator step.
        noise_rtg [S,N,3,R] / noise_act [S,N,V] float32 tensors (explicit Exp(1) noise) or None (in-kernel)."""
        self.policy_step(t, noise_rtg, noise_act)
        self.sim_step(t)

    def sim_step(self, t, act_f64=None, s0=0, s1=None):
        """Simulator step of scenarios [s0, s1) (default: all)."""
        lib, p, st = self.lib, _lib.ptr, _lib.stream_ptr()
        s1 = self.S if s1 is None else s1
        sl = slice(s0, s1)
        _lib.check(lib.ctrlsim_sim_step(s1 - s0, self.N, self.E, p(self.act_now[sl]) if act_f64 is None else None,
                                        p(act_f64[sl]) if act_f64 is not None else None, self.disc6, p(self.size[sl]),
                                        p(self.edges[sl]), p(self.exists[sl]), p(self.phys[sl]), p(self.hist_states[sl]),
                                        p(self.coll[sl]), None, t, self.steps + 1, self.dt, self.kinematic,
                                        p(self.contact_state[sl]) if self.contact_state is not None else None, st), "sim_step")

    def _group_build(self, t, s0=0, s1=None):
        lib, p, st, d = self.lib, _lib.ptr, _lib.stream_ptr(), self.dims
        s1 = self.S if s1 is None else s1
        sl = slice(s0, s1)
        _lib.check(lib.ctrlsim_group_build(s1 - s0, self.N, d.A, d.T, t, self.steps + 1, float(self.w.agent_dist_threshold),
                                           p(self.hist_states[sl]), p(self.eval_order[sl]), 1 if self.P_all > 0 else 0,
                                           p(self.persist[sl]), p(self.n_groups[sl]), p(self.grp_focal[sl]),
                                           p(self.grp_ids[sl]), p(self.grp_members[sl]), p(self.own_g[sl]), p(self.mem_g[sl]),
                                           p(self.tilted[sl]), st), "group_build")

    # ------------------------------------------------------------------ cached phase (t < T), chunk-major
    def _chunk_step_cached(self, s0, s1, B, t, ws):
        """Policy + simulator step of one chunk of scenarios with the decoder K/V cache of that chunk (`ws`)."""
        lib, p, st, d = self.lib, _lib.ptr, _lib.stream_ptr(), self.dims
        N, Tmax, ns, sl = self.N, self.steps, s1 - s0, slice(s0, s1)
        Tq, tt_first = t + 1, max(t - 1, 0)
        _lib.check(lib.ctrlsim_ctx_index(s0, s1, N, p(self.n_groups), p(self.grp_focal), p(self.grp_ids), p(self.own_g),
                                         p(self.mem_g), p(self.ctx_scn), p(self.ctx_grp), p(self.own_ctx), p(self.own_slot),
                                         p(self.mem_ctx), p(self.mem_slot), p(self.ctx_base), st), "ctx_index")
        _lib.check(lib.ctrlsim_build_context(B, N, d.A, d.T, t, Tq, tt_first, Tmax + 1, Tmax, self.P_all, d.P, d.NP,
                                             p(self.ctx_scn), p(self.ctx_grp), p(self.grp_focal), p(self.grp_ids),
                                             p(self.hist_states), p(self.hist_tok), p(self.hist_rtg), p(self.goals),
                                             p(self.types), p(self.roads), p(self.rtypes), self._zero4,
                                             C.byref(self.ctx.struct), st), "build_context")
        _lib.check(lib.ctrlsim_dt_forward_pass1_cached(self.model.handle, B, t, C.byref(self.ctx.struct), p(ws),
                                                       p(self.rtg_logits), st), "pass1_cached")
        _lib.check(lib.ctrlsim_sample_rtg(p(self.rtg_logits), d.A, d.R, p(self.own_ctx[sl]), p(self.own_slot[sl]),
                                          p(self.tilted[sl]), self.tilt,
                                          p(self.tilt_scn[sl]) if self.tilt_scn is not None else None, None, self.seed,
                                          p(self.scenario_id[sl]), t, p(self.hist_rtg[sl]), ns, N, Tmax, st), "sample_rtg")
        _lib.check(lib.ctrlsim_dt_forward_pass2(self.model.handle, B, Tq, t, N, Tmax, C.byref(self.ctx.struct), p(self.ctx_scn),
                                                p(self.hist_rtg), p(ws), p(self.act_logits), 1, st), "pass2_cached")
        _lib.check(lib.ctrlsim_sample_action(p(self.act_logits), d.A, d.V, p(self.mem_ctx[sl]), p(self.mem_slot[sl]),
                                             self.temperature, self.top_p, None, self.seed, p(self.scenario_id[sl]), t,
                                             p(self.hist_tok[sl]), p(self.act_now[sl]), ns, N, Tmax, ZERO_ACTION_TOKEN, st),
                   "sample_action")
        self.sim_step(t, s0=s0, s1=s1)

    def _run_cached_phase(self, n_steps):
        """Steps 0 .. n_steps-1 (n_steps <= T) chunk by chunk: while t < T the window starts at step 0, so a context's
        frame, membership and map are constant and its decoder K/V can be cached across steps (csrc/forward.hip).
        A chunk whose context set does change (a vehicle stops existing) falls back to the full recompute."""
        self._group_build(0)
        self.n_groups_host.copy_(self.n_groups, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        counts = self.n_groups_host.numpy().copy()
        self.groups_per_step[0] = counts
        for (s0, s1, B) in self._chunks(counts):
            sl = slice(s0, s1)
            cached_ok = B > 0
            ref_focal, ref_ids = self.grp_focal[sl].clone(), self.grp_ids[sl].clone()
            for t in range(n_steps):
                if t > 0:
                    self._group_build(t, s0, s1)
                    self.n_groups_host[sl].copy_(self.n_groups[sl], non_blocking=True)
                    same = torch.equal(self.grp_focal[sl], ref_focal) and torch.equal(self.grp_ids[sl], ref_ids)   # syncs
                    cnt = self.n_groups_host.numpy()[sl]
                    self.groups_per_step[t, sl] = cnt
                    cached_ok = cached_ok and same and np.array_equal(cnt, counts[sl])
                if cached_ok:
                    self._chunk_step_cached(s0, s1, B, t, self.ws)
                else:
                    self._policy_chunks(t, self.n_groups_host.numpy(), s0, s1)
                    self.sim_step(t, s0=s0, s1=s1)

    def policy_step(self, t, noise_rtg=None, noise_act=None):
        """AutoregressivePolicy.predict for every scenario: writes hist_rtg[..., t, :], hist_tok[..., t], act_now."""
        self._group_build(t)
        self.n_groups_host.copy_(self.n_groups, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        counts = self.n_groups_host.numpy()
        self.groups_per_step[t] = counts
        self._policy_chunks(t, counts, 0, self.S, noise_rtg, noise_act)

    def _policy_chunks(self, t, counts, lo, hi, noise_rtg=None, noise_act=None):
        """Full-recompute policy for scenarios [lo, hi), chunked to the model batch."""
        lib, p, st, d = self.lib, _lib.ptr, _lib.stream_ptr(), self.dims
        S, N, Tmax = self.S, self.N, self.steps
        Tq = min(t, d.T - 1) + 1
        for (s0, s1, B) in [(lo + a, lo + b, c) for (a, b, c) in self._chunks(counts[lo:hi])]:
            ns = s1 - s0
            sl = slice(s0, s1)
            if B > 0:
                _lib.check(lib.ctrlsim_ctx_index(s0, s1, N, p(self.n_groups), p(self.grp_focal), p(self.grp_ids),
                                                 p(self.own_g), p(self.mem_g), p(self.ctx_scn), p(self.ctx_grp),
                                                 p(self.own_ctx), p(self.own_slot), p(self.mem_ctx), p(self.mem_slot),
                                                 p(self.ctx_base), st), "ctx_index")
                _lib.check(lib.ctrlsim_build_context(B, N, d.A, d.T, t, Tq, 0, Tmax + 1, Tmax, self.P_all, d.P, d.NP,
                                                     p(self.ctx_scn), p(self.ctx_grp), p(self.grp_focal), p(self.grp_ids),
                                                     p(self.hist_states), p(self.hist_tok), p(self.hist_rtg),
                                                     p(self.goals), p(self.types), p(self.roads), p(self.rtypes),
                                                     self._zero4, C.byref(self.ctx.struct), st), "build_context")
                if d.VARIANT:                                # IL / Trajeglish: no RTG tokens, one forward (predict_rtgs False)
                    _lib.check(lib.ctrlsim_dt_forward_actions(self.model.handle, B, Tq, C.byref(self.ctx.struct), p(self.ws),
                                                              p(self.act_logits), st), "forward_actions")
                else:
                    _lib.check(lib.ctrlsim_dt_forward_pass1(self.model.handle, B, Tq, C.byref(self.ctx.struct), p(self.ws),
                                                            p(self.rtg_logits), None, st), "pass1")
            else:
                self.own_ctx[sl].fill_(-1)
                self.mem_ctx[sl].fill_(-1)
            if not d.VARIANT:
                _lib.check(lib.ctrlsim_sample_rtg(p(self.rtg_logits), d.A, d.R, p(self.own_ctx[sl]), p(self.own_slot[sl]),
                                                  p(self.tilted[sl]), self.tilt,
                                                  p(self.tilt_scn[sl]) if self.tilt_scn is not None else None,
                                                  p(noise_rtg[sl]) if noise_rtg is not None else None, self.seed,
                                                  p(self.scenario_id[sl]), t, p(self.hist_rtg[sl]), ns, N, Tmax, st),
                           "sample_rtg")
            if B > 0 and not d.VARIANT:
                _lib.check(lib.ctrlsim_dt_forward_pass2(self.model.handle, B, Tq, t, N, Tmax, C.byref(self.ctx.struct),
                                                        p(self.ctx_scn), p(self.hist_rtg), p(self.ws),
                                                        p(self.act_logits), 0, st), "pass2")
            _lib.check(lib.ctrlsim_sample_action(p(self.act_logits), d.A, d.V, p(self.mem_ctx[sl]), p(self.mem_slot[sl]),
                                                 self.temperature, self.top_p,
                                                 p(noise_act[sl]) if noise_act is not None else None, self.seed,
                                                 p(self.scenario_id[sl]), t, p(self.hist_tok[sl]), p(self.act_now[sl]),
                                                 ns, N, Tmax, ZERO_ACTION_TOKEN, st), "sample_action")

    def run(self, steps=None, noise_fn=None):
        """Roll all loaded scenarios `steps` steps.  noise_fn(t) -> (noise_rtg, noise_act) or None."""
        steps = self.steps if steps is None else steps
        if self.dims.VARIANT == 3:
            raise NotImplementedError("the Decision-Transformer policy conditions on real-time rewards computed by the rollout "
                                      "driver: run it through PolicyEvaluator (hist_rtg is fed per step), not RolloutEngine.run")
        start = 0
        if self.use_cache and noise_fn is None and steps > 0:
            start = min(self.dims.T, steps)
            self._run_cached_phase(start)
        for t in range(start, steps):
            if noise_fn is not None:
                nr, na = noise_fn(t)
                self.step(t, nr, na)
            else:
                self.step(t)
        return self

    def results(self):
        torch.cuda.synchronize(self.device)
        bad = int(self.lib.ctrlsim_nonfinite_count(1))
        if bad:
            raise FloatingPointError(f"{bad} sampling races had no finite logit (NaN in the forward pass: an activation beyond "
                                     "the fp16 range of the split operands, csrc/split.h, or bad weights)")
        return dict(tokens=self.hist_tok.cpu().numpy(), rtg_bins=self.hist_rtg.cpu().numpy(),
                    states=self.hist_states.cpu().numpy(), coll=self.coll.cpu().numpy(),
                    n_groups=self.groups_per_step.copy())
