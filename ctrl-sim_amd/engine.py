"""Batched closed-loop rollout engine: S independent scenarios resident in HBM, stepped by HIP kernels.

This is the MI355X-native replacement of the reference's per-scenario Python loop
(evaluators/policy_evaluator.py:514-557: update dict -> Policy.update_state -> AutoregressivePolicy.predict ->
act -> Simulation.step).  All per-step state lives on the device ([S, N, ...] arrays, see include/ctrlsim.h); the host
only sequences kernel launches through the C ABI and reads ONE small array per lane and step (groups per scenario, needed
to size the model batch).  torch is used for device memory, streams and events only.

Scenarios in flight are sharded over LANES (the intra-GPU "one scenario set per stream" sharding of the north star):
a lane = a contiguous scenario range with its own model workspace, context tensors and side HIP stream.  The matrix
kernels of all lanes run back to back on ONE main stream (they each fill the chip; keeping them in order keeps their
per-launch timing clean), while a lane's simulator step, the focal grouping of its next step and the small
device -> host copy of the group counts run on the lane's side stream, concurrently with the other lane's forward
pass.  The host waits for a lane's counts only after it has queued the other lane's whole step, so the GPU never
idles on that round trip: with two lanes the per-step host synchronisation, the single-lane-per-scenario solver of
`sim_step` and the grouping kernels disappear from the critical path.

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
import os

import numpy as np
import torch

from . import _lib, pack as _pack
from .spec import Dims, ZERO_ACTION_TOKEN, ZERO_RTG_BINS, check_supported


def _dims_struct(d: Dims):
    return _lib.Dims(A=d.A, T=d.T, P=d.P, NP=d.NP, D=d.D, H=d.H, F=d.F, V=d.V, R=d.R, C=d.C, NE=d.NE, ND=d.ND,
                     MAXT=d.MAXT, variant=4 if getattr(d, "MASK_OWN", False) else d.VARIANT,    # 4: CtRL-Sim tokens, own-return mask
                     flags=int(getattr(d, "FLAGS", 0)))


class HipModel:
    """Device-resident weights + the opaque ctrlsim_model handle."""

    def __init__(self, cfg, weights: dict, device="cuda:0"):
        self.cfg = cfg
        check_supported(cfg)                    # a cfg of another network (mask / token / map options) is refused by name
        self.dims = Dims(cfg)
        self.device = torch.device(device)
        self.lib = _lib.lib()
        if self.dims.FLAGS & 2:
            # cfg.model.use_map = False: the checkpoint has no encoder.map_encoder.* (modules/encoder.py:18).  The kernels still run the map
            # side — its polyline rows are key-padded everywhere (ctrlsim_dims.flags bit 1) — on zeros of the shapes the module would have
            from . import weights as _w
            weights = dict(weights)
            for k, v in _w.generate(self.dims, 0).items():
                if "map_encoder." in k and k not in weights:
                    weights[k] = np.zeros_like(v)
        flat, names, offsets = _pack.pack(self.dims, weights)
        self.flat = torch.from_numpy(flat).to(self.device)
        self._names = (C.c_char_p * len(names))(*[n.encode() for n in names])
        self._offsets = (C.c_int64 * len(names))(*[int(o) for o in offsets])
        self.cdims = _dims_struct(self.dims)
        h = C.c_void_p()
        _lib.check(self.lib.ctrlsim_model_create(C.byref(self.cdims), self.flat.data_ptr(), len(names), self._names,
                                                 self._offsets, C.byref(h)), "model_create")
        self.handle = h
        self.split_fallback = False     # an engine of this model met non-finite values under f16x3: later split="auto" engines
                                        # (the policy surface opens one per scenario session) start on bf16x6 right away

    def workspace_bytes(self, B, Tq, A=None):
        n = self.lib.ctrlsim_forward_workspace_bytes_a(C.byref(self.cdims), B, Tq, self.dims.A if A is None else A)
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
        self.struct = _lib.Ctx(*(getattr(self, k).data_ptr() for k in self.FIELDS))
        self._d = d

    FIELDS = ("st12", "exist", "goal5", "act_tok", "rtg_bin", "tstep", "slot_gid", "road_pts", "road_types")

    def class_structs(self, classes, Tn):
        """One ctrlsim_ctx per size class: the buffers are shared, class k's B_k contexts of A_k slots x Tn window rows start
        where the previous classes' end.  classes = [(B_k, A_k)]."""
        d = self._d
        per_ctx = lambda A: dict(st12=Tn * A * 12, exist=Tn * A, goal5=A * 5, act_tok=Tn * A, rtg_bin=Tn * A * 3, tstep=Tn,
                                 slot_gid=A, road_pts=d.P * d.NP * 3, road_types=d.P * 8)
        off = {k: 0 for k in self.FIELDS}
        out = []
        for B, A in classes:
            out.append(_lib.Ctx(*(getattr(self, k).data_ptr() + 4 * off[k] for k in self.FIELDS)))
            n = per_ctx(A)
            for k in self.FIELDS:
                off[k] += B * n[k]
        return out


def ctx_from_reference_layout(d: Dims, data: dict, Tq: int, device):
    """Reference-layout model inputs (agent_states [B,A,T,8], ... as in modules/encoder.py:52-63) -> CtxBuffers with
    the first Tq window steps.  The arrays may hold fewer slots than d.A (a compact context: the leading A' slots of the
    reference layout, the last one a padded slot).  Used by the model-level parity tests."""
    B, A = data["agent_states"].shape[:2]
    cb = CtxBuffers(d, B, device)
    st = np.asarray(data["agent_states"], np.float64)
    types = np.broadcast_to(np.asarray(data["agent_types"], np.float64)[:, :, None, :], (B, A, d.T, 5))
    st12 = np.concatenate([st[..., :7], types], -1).transpose(0, 2, 1, 3)[:, :Tq]
    flat = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(device).reshape(-1)
    cb.st12.view(-1)[:B * Tq * A * 12] = flat(st12, np.float32)
    cb.exist.view(-1)[:B * Tq * A] = flat(st[..., 7].transpose(0, 2, 1)[:, :Tq], np.float32)
    cb.goal5.view(-1)[:B * A * 5] = flat(np.asarray(data["goals"], np.float64), np.float32)
    cb.act_tok.view(-1)[:B * Tq * A] = flat(np.asarray(data["actions"]).transpose(0, 2, 1)[:, :Tq], np.int32)
    rt = np.asarray(data["rtgs"]).transpose(0, 2, 1, 3)[:, :Tq]
    if d.VARIANT == 3:                                   # decision transformer: continuous RTGs travel as float bits
        rt = np.ascontiguousarray(rt, np.float32).view(np.int32)
    cb.rtg_bin.view(-1)[:B * Tq * A * 3] = flat(rt, np.int32)
    cb.tstep.view(-1)[:B * Tq] = flat(np.asarray(data["timesteps"])[:, 0, :Tq, 0], np.int32)
    cb.slot_gid.fill_(-1)
    cb.road_pts.copy_(torch.from_numpy(np.asarray(data["road_points"], np.float64).astype(np.float32)).to(device))
    cb.road_types.copy_(torch.from_numpy(np.asarray(data["road_types"], np.float64).astype(np.float32)).to(device))
    return cb


def _new_stream(dev, cu_mask=None):
    """A stream of its own; cu_mask (A/B experiments: "w0,w1,..." 32-bit hex words, bit i = compute unit i enabled) creates it with
    hipExtStreamCreateWithCUMask so that its kernels run on those compute units only (CU partition experiment)."""
    if not cu_mask:
        prio = os.environ.get("CTRLSIM_SIDE_PRIORITY")           # A/B switch: queue priority of the side streams (-1 = high)
        return torch.cuda.Stream(device=dev, priority=int(prio)) if prio else torch.cuda.Stream(device=dev)
    words = [int(w, 16) for w in cu_mask.split(",")]
    arr = (C.c_uint32 * len(words))(*words)
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(st.value, device=dev)


class _Lane:
    """A scenario set in flight: model workspace (K/V cache included), context tensors, logits, chunk index lists, a side
    stream for its simulator step / grouping / count read-back, and the two events that order it against the main stream."""

    def __init__(self, eng, idx, own_stream):
        d, dev = eng.dims, eng.device
        B = eng.ctx_cap                                           # contexts per model batch (compact ones are cheap: see _chunks)
        self.idx = idx
        self.ctx = CtxBuffers(d, B, dev)
        self.ws = torch.empty(eng._ws_bytes + (1 << 20), dtype=torch.uint8, device=dev)   # max_ctx plain contexts always fit,
                                                                                            # whatever the class mix and the split
        self.rtg_logits = torch.empty(B, d.A, d.R * d.C, device=dev)
        self.act_logits = torch.empty(B, d.A, d.V, device=dev)
        self.ctx_scn = torch.zeros(B, dtype=torch.int32, device=dev)
        self.ctx_grp = torch.zeros(B, dtype=torch.int32, device=dev)
        self.ctx_row0 = torch.zeros(B, dtype=torch.int32, device=dev)
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.host_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.host_hist = None                                    # pinned [S, classes] int32, sized by load_scenarios
        self.side = _new_stream(dev, os.environ.get("CTRLSIM_SIDE_CU_MASK")) if own_stream else None
        self.ev_fwd, self.ev_ready, self.ev_p2, self.ev_p1, self.ev_sim, self.ev_ctx = (torch.cuda.Event() for _ in range(6))
        self.sim_in_flight = False                               # a simulator step of this lane may still run on its side stream
        self.pending = (0, 0, False)                             # scenario range (+ compare flag) of the read-back in flight


class RolloutEngine:
    def __init__(self, cfg, weights: dict, device="cuda:0", max_ctx=256, seed=0, tilt=(0.0, 0.0, 0.0),
                 temperature=None, nucleus=None, top_p=None, kinematic=False, model=None, use_cache=True, contacts=True,
                 lanes=1, compact=True, split="auto", sizes=None, options=None):
        self.cfg = cfg
        self.w = cfg.dataset.waymo
        self.dims = Dims(cfg)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.model = model if model is not None else HipModel(cfg, weights, device)
        self.lib = self.model.lib
        # operand split of the matrix kernels (csrc/split.h; the option is process-global): "f16x3" two fp16 planes, three products;
        # "bf16x6" three bf16 planes, six products — half the speed, the whole fp32 exponent range; "auto" = f16x3, and a rollout
        # whose sampling met non-finite logits (an operand beyond the fp16 range: trained weights can do that) is repeated with bf16x6
        assert split in ("auto", "f16x3", "bf16x6")
        self.split = split
        # Per-ENGINE library state (ctrlsim_bind, re-asserted at the top of every run / step): the operand split this engine's K/V
        # images and workspace are used with, and its own guard counter (non-finite LayerNorm rows / sampling races, simulator
        # contact-table overflows) — several engines (planner and adversary policies, a second model) take turns in one process
        # without changing each other's kernels or reading each other's events.
        # guard pair (include/ctrlsim.h: ctrlsim_bind): [0] non-finite events of the model, [1] simulator events — separate words, so
        # that a real fp16 overflow at production scale (thousands of LayerNorm rows per step) cannot carry into the simulator's count
        self.guard = torch.zeros(2, dtype=torch.int32, device=self.device)
        # this engine's kernel options (include/ctrlsim.h: ctrlsim_bind_options; {key: value}, keys of ctrlsim_set_option): entries
        # override the process defaults for this engine's launches only — two engines with different options take turns in one process
        self._options = (C.c_int * int(self.lib.ctrlsim_option_count()))(*([-1] * int(self.lib.ctrlsim_option_count())))
        for k, v in (options or {}).items():
            self._check_option_key(k)
            self._options[int(k)] = int(v)
        self.scheme = 0 if (split == "bf16x6" or (split == "auto" and self.model.split_fallback)) else 1
        self._unchecked = []                    # fresh-from-reset runs since the last check_finite: (steps, s0, s1)
        self._bind()
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
        # the K/V-cached phase is built for the CtRL-Sim tokens under the default mask (attend_own_return_action: full recompute every step)
        self.use_cache = bool(use_cache) and not self.dims.VARIANT and not self.dims.MASK_OWN
        # Few-row kernels on the lanes' side streams (second pass, the tail of the first pass, whole K/V-cached steps) underneath the
        # other lane's full-row kernels: +6.5 % throughput (113.4 -> 120.7 k agent-steps/s).  ON by default since round 3: the
        # irreproducibility that kept them off in round 2 was traced to code of clang's SLP vectoriser (packed-fp32 instructions with
        # operand swizzles) in workgroups that share a CU with another kernel's matrix-pipe workgroups; the library is built without
        # that pass (csrc/build.py) and provoked rollouts are bit-identical to single-stream ones (DESIGN.md section 4,
        # tests/test_gpu_hazard.py).  forward_waits_for_sim is the round-2 stream guard (a forward pass waits for every pending
        # simulator step): not needed any more, and it serialises exactly the overlap the switches create; kept as an A/B switch.
        self.forward_waits_for_sim = False
        # context build on the lane's side stream (round 4 experiment: 129.44 -> 129.73 k, inside the noise — off by default)
        self.ctx_on_side = os.environ.get("CTRLSIM_CTX_ON_SIDE", "0") == "1"
        self._max_sliding, self._sliding, self._holds_slot = 0, 0, []
        self.full_pass_contexts = 0             # np.int64 [classes] once the first full pass ran
        self.record_phases = False              # bench.py: main-stream events at the end of every lane's K/V-cached phase
        self.phase_events = []
        self.pass2_on_side = True
        self.tail_on_side = True
        self.cached_on_side = True
        self.device_ledger = self.dims.VARIANT == 3    # DT: RTG rows from the device reward ledger (the plugin surface feeds hist_rtg itself)
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
        # size classes of compact contexts (include/ctrlsim.h: ctrlsim_group_size_hist): slot counts 4, 8, ... and A.  A context
        # with n vehicles runs with the first size >= n + 1; the CtRL-Sim model only (the baselines keep the plain layout).
        A = self.dims.A
        # The 24-slot set is fitted to the occupancy of the bench's scenes (tools/microbench/nstat.py: mean 9.3 vehicles per
        # context in the sliding-window phase, 92 % of the contexts hold 3..15): with the 16 classes a launch's class tables
        # hold (CTRLSIM_MAX_CLASSES) 1.014x the rows and 1.033x the attention pairs of exact per-context sizes — the eight classes
        # of round 2 (6, 8, 10, 12, 14, 16, 20, 24) cost 1.080x / 1.142x.  Row-wise kernels run once over all classes and the
        # attention grid is a concatenation, so a class costs nothing but a few index launches.
        # Other slot counts (tests, the non-reference wide context A = 64): every size up to 17 slots, else 16 sizes spread evenly up to A.
        tuned = (4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 24) if A == 24 else \
            tuple(range(2, A + 1))[-16:] if A <= 17 else tuple(sorted({max(2, (A * k + 15) // 16) for k in range(1, 17)}))
        if sizes is not None:                       # explicit class set (A/B measurements, tests): ascending slot counts, A last
            tuned = tuple(int(a) for a in sizes)
            assert tuned == tuple(sorted(set(tuned))) and tuned[-1] == A and tuned[0] >= 2 and len(tuned) <= 16
        # (compact contexts rest on the default mask's treatment of padded slots: attend_own_return_action keeps the plain 24-slot layout)
        self.sizes = tuned if (compact and not self.dims.VARIANT and not self.dims.MASK_OWN) else (A,)
        self._sizes_c = (C.c_int * len(self.sizes))(*self.sizes)
        self.ctx_cap = self.max_ctx * (4 if len(self.sizes) > 1 else 1)
        # workspace bytes per context of each class (the carve is linear in B up to alignment), + a fixed allowance per class
        if self.split == "auto":
            self.lib.ctrlsim_bind(0, self.guard.data_ptr())    # size the workspace for the larger (three-plane) K/V images
        wb = self.model.workspace_bytes
        self._bpc = [(wb(257, self.dims.T, a) - wb(1, self.dims.T, a)) / 256.0 for a in self.sizes]
        self._ws_fixed = sum(wb(1, self.dims.T, a) for a in self.sizes) + (1 << 20)
        self._ws_bytes = wb(self.max_ctx, self.dims.T) + self._ws_fixed
        self._bind()
        self.n_lanes = max(1, int(lanes))
        self.lanes = [_Lane(self, i, self.n_lanes > 1) for i in range(self.n_lanes)]
        L0 = self.lanes[0]                   # the synchronous single-stream entry points (policy_step / step) use lane 0
        self.ctx, self.ws, self.rtg_logits, self.act_logits = L0.ctx, L0.ws, L0.rtg_logits, L0.act_logits
        self.ctx_scn, self.ctx_grp = L0.ctx_scn, L0.ctx_grp
        self._ws_cap = L0.ws.numel()
        self._main = torch.cuda.current_stream(self.device)
        self.S = 0

    def _bind(self):
        _lib.check(self.lib.ctrlsim_bind(int(self.scheme), self.guard.data_ptr()), "bind")
        _lib.check(self.lib.ctrlsim_bind_options(self._options), "bind_options")

    _OPT_SPLIT = 4                              # include/ctrlsim.h: the operand split is bound with the images it was built for

    def _check_option_key(self, key):
        if int(key) == self._OPT_SPLIT:
            raise ValueError("option 4 (operand split) is not an engine option: the weight planes, K/V images and workspace of an engine are "
                             "laid out for ONE split — choose it with RolloutEngine(split='f16x3' | 'bf16x6' | 'auto')")
        if not 0 <= int(key) < len(self._options):
            raise ValueError(f"unknown kernel option {key} (include/ctrlsim.h: 0 .. {len(self._options) - 1})")

    def set_option(self, key, value):
        """Kernel option `key` (include/ctrlsim.h: ctrlsim_set_option) for THIS engine's launches; value None = inherit the process default."""
        self._check_option_key(key)
        self._options[int(key)] = -1 if value is None else int(value)
        self._bind()

    def __del__(self):
        try:                                    # the library must not keep a pointer into memory torch is about to recycle
            self.lib.ctrlsim_unbind(self.guard.data_ptr())
        except Exception:
            pass

    def _set_split(self, scheme):
        self.scheme = int(scheme)
        if self.split == "auto" and self.scheme == 0:
            self.model.split_fallback = True
        self._bind()

    def nonfinite(self, reset=True):
        """Synchronise and read this engine's guard counter (non-finite LayerNorm rows / sampling races, contact-table overflows)."""
        torch.cuda.synchronize(self.device)
        nf, sim = (int(v) for v in self.guard.tolist())
        if (nf or sim) and reset:
            self.guard.zero_()
        # the legacy encoding of ctrlsim_nonfinite_count: low half = non-finite events of the model, high half = simulator events, both
        # SATURATED — the device counts them in separate words, neither can carry into the other
        return min(nf, 65535) + 65536 * min(sim, 32767)

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
        if self.dims.VARIANT == 3:      # Decision Transformer: real-time reward ledger on the device (csrc/rewards.hip)
            self.dt_ledger = z(S, N, 10, dt=torch.float64)
            self.dt_rtg_raw = z(S, N, Tmax, 3, dt=torch.float64)
            self.dt_init_rtg = None                          # [S,N,3] float64 tensor, or None = (10, 90, 90) (max_return)
        self.act_now = z(S, N, dt=torch.int32)
        self.applied = z(S, N, Tmax, 2, dt=torch.float64)
        self.persist = z(S, N, dt=torch.int64)
        self.n_groups = z(S, dt=torch.int32)
        nb = len(self.sizes)
        self.size_hist = z(S, nb, dt=torch.int32)
        self.ctx_of_group = z(S, N, dt=torch.int32)
        for L in self.lanes:
            L.host_hist = torch.zeros(S, nb, dtype=torch.int32).pin_memory()
        self.grp_focal = z(S, N, dt=torch.int32)
        self.grp_ids = z(S, N, dt=torch.int64)
        self.grp_members = z(S, N, dt=torch.int64)
        # snapshot of the focal groups a chunk's K/V cache was built for (ctrlsim_groups_changed)
        self.ref_n, self.ref_focal, self.ref_ids = z(S, dt=torch.int32), z(S, N, dt=torch.int32), z(S, N, dt=torch.int64)
        self.own_g, self.mem_g = z(S, N, dt=torch.int32), z(S, N, dt=torch.int32)
        self.tilted = z(S, N, dt=torch.uint8)
        self.own_ctx, self.own_slot = z(S, N, dt=torch.int32), z(S, N, dt=torch.int32)
        self.mem_ctx, self.mem_slot = z(S, N, dt=torch.int32), z(S, N, dt=torch.int32)
        self.ctx_base = z(S, dt=torch.int32)
        self._zero_rtg_row = torch.tensor(self.zero_rtg, dtype=torch.int32, device=dev)
        self.groups_per_step = np.zeros((self.steps, S), np.int32)
        self.guard.zero_()
        self._unchecked = []
        self.reset()

    def reset(self, s0=0, s1=None):
        """Back to step 0 for scenarios [s0, s1) (default: all)."""
        st = _lib.stream_ptr()
        s1 = self.S if s1 is None else s1
        sl = slice(s0, s1)
        self._fresh = (s0, s1)
        self.hist_states[sl].zero_()
        self.coll[sl].zero_()
        self.hist_tok[sl].fill_(ZERO_ACTION_TOKEN)
        self.hist_rtg[sl].copy_(self._zero_rtg_row.expand_as(self.hist_rtg[sl]))
        self.persist[sl].zero_()
        self.applied[sl].zero_()
        p = _lib.ptr
        _lib.check(self.lib.ctrlsim_sim_init(s1 - s0, self.N, self.E, p(self.init_pose[sl]), p(self.size[sl]), p(self.edges[sl]),
                                             p(self.exists[sl]), p(self.phys[sl]), p(self.hist_states[sl]), p(self.coll[sl]),
                                             self.steps + 1, p(self.contact_state[sl]) if self.contact_state is not None else None,
                                             st), "sim_init")

    # ------------------------------------------------------------------ chunk plan
    def _chunks(self, hist, base=0):
        """Cut scenarios base .. base+len(hist) into model batches.  hist [n, classes] = focal groups per scenario and size
        class.  A batch must fit the lane's workspace (compact contexts are cheaper: bytes per context by class) and its
        context buffers (ctx_cap contexts); batches are balanced (a short last batch costs a whole set of launches).
        -> [(s0, s1, counts_per_class)]."""
        hist = np.asarray(hist, np.int64).reshape(-1, len(self.sizes))
        cost = hist @ np.asarray(self._bpc)                       # workspace bytes per scenario
        nctx = hist.sum(1)
        cap_b = self._ws_cap - self._ws_fixed - (1 << 20)
        cap_c = self.ctx_cap
        if len(hist) and (cost.max() > cap_b or nctx.max() > cap_c):
            raise RuntimeError(f"a scenario has {int(nctx.max())} focal groups: more than one model batch (max_ctx={self.max_ctx})")
        n = max(1, int(np.ceil(max(cost.sum() / cap_b, nctx.sum() / cap_c))))
        lim_b = min(cap_b, cost.sum() / n + (cost.max() if len(cost) else 0))
        lim_c = min(cap_c, nctx.sum() / n + (nctx.max() if len(nctx) else 0))
        chunks, s0, acc_b, acc_c = [], 0, 0.0, 0
        for s in range(len(hist)):
            if acc_b + cost[s] > lim_b or acc_c + nctx[s] > lim_c or s - s0 >= 4095:
                chunks.append((base + s0, base + s, hist[s0:s].sum(0)))
                s0, acc_b, acc_c = s, 0.0, 0
            acc_b += cost[s]
            acc_c += nctx[s]
        chunks.append((base + s0, base + len(hist), hist[s0:].sum(0)))
        return [(a, b, [int(x) for x in c]) for (a, b, c) in chunks if b > a]

    def _class_plan(self, L, counts, Tw, Tn):
        """The non-empty size classes of a model batch as the arguments of the batch entry points: ([(B_k, A_k, first context,
        ctx struct)], n, B[] (c_int array), A[], ctx[] (ctrlsim_ctx array)).  Tw = window steps the workspace is carved for,
        Tn = window rows held by the context tensors."""
        classes = [(B, A) for B, A in zip(counts, self.sizes)]
        structs = L.ctx.class_structs(classes, Tn)
        plan, c0 = [], 0
        for k, (B, A) in enumerate(classes):
            if B > 0:
                plan.append((B, A, c0, structs[k]))
            c0 += B
        n = len(plan)
        Bs = (C.c_int * max(n, 1))(*[q[0] for q in plan])
        As = (C.c_int * max(n, 1))(*[q[1] for q in plan])
        cs = (_lib.Ctx * max(n, 1))(*[q[3] for q in plan])
        if n:
            need = self.lib.ctrlsim_forward_workspace_bytes_c(C.byref(self.model.cdims), n, Bs, As, Tw)
            assert 0 < need <= self._ws_cap, "model batch exceeds the lane workspace"
        return plan, n, Bs, As, cs

    # ------------------------------------------------------------------ kernels of one step, on explicit streams
    def sim_step(self, t, act_f64=None, s0=0, s1=None, stream=None):
        """Simulator step of scenarios [s0, s1) (default: all)."""
        lib, p = self.lib, _lib.ptr
        if stream is None:
            self._bind()                        # direct calls (plugin surface); run() binds once for its whole schedule
        st = _lib.stream_ptr() if stream is None else stream
        s1 = self.S if s1 is None else s1
        sl = slice(s0, s1)
        _lib.check(lib.ctrlsim_sim_step(s1 - s0, self.N, self.E, p(self.act_now[sl]) if act_f64 is None else None,
                                        p(act_f64[sl]) if act_f64 is not None else None, self.disc6, p(self.size[sl]),
                                        p(self.edges[sl]), p(self.exists[sl]), p(self.phys[sl]), p(self.hist_states[sl]),
                                        p(self.coll[sl]), None, t, self.steps + 1, self.dt, self.kinematic,
                                        p(self.contact_state[sl]) if self.contact_state is not None else None, st), "sim_step")

    def _group_build(self, t, s0=0, s1=None, stream=None):
        lib, p, d = self.lib, _lib.ptr, self.dims
        st = _lib.stream_ptr() if stream is None else stream
        s1 = self.S if s1 is None else s1
        sl = slice(s0, s1)
        _lib.check(lib.ctrlsim_group_build(s1 - s0, self.N, d.A, d.T, t, self.steps + 1, float(self.w.agent_dist_threshold),
                                           p(self.hist_states[sl]), p(self.eval_order[sl]), 1 if self.P_all > 0 else 0,
                                           p(self.persist[sl]), p(self.n_groups[sl]), p(self.grp_focal[sl]),
                                           p(self.grp_ids[sl]), p(self.grp_members[sl]), p(self.own_g[sl]), p(self.mem_g[sl]),
                                           p(self.tilted[sl]), st), "group_build")

    def _side(self, L):
        return L.side if L.side is not None else self._main

    def _enqueue_groups(self, L, t, s0, s1, compare=False):
        """Side stream of lane L: focal groups of step t for scenarios [s0, s1), (compare) the changed-vs-snapshot flag, the
        asynchronous read-back of the group counts per size class (+ flag), then L.ev_ready."""
        side = self._side(L)
        sl = slice(s0, s1)
        self._group_build(t, s0, s1, side.cuda_stream)
        with torch.cuda.stream(side):
            if compare:
                L.flag.zero_()
                p = _lib.ptr
                _lib.check(self.lib.ctrlsim_groups_changed(s1 - s0, self.N, p(self.n_groups[sl]), p(self.grp_focal[sl]),
                                                           p(self.grp_ids[sl]), p(self.ref_n[sl]), p(self.ref_focal[sl]),
                                                           p(self.ref_ids[sl]), p(L.flag), side.cuda_stream), "groups_changed")
                L.host_flag.copy_(L.flag, non_blocking=True)
            _lib.check(self.lib.ctrlsim_group_size_hist(s1 - s0, self.N, _lib.ptr(self.n_groups[sl]), _lib.ptr(self.grp_ids[sl]),
                                                        len(self.sizes), self._sizes_c, _lib.ptr(self.size_hist[sl]),
                                                        side.cuda_stream), "group_size_hist")
            L.host_hist[:s1 - s0].copy_(self.size_hist[sl], non_blocking=True)
            L.ev_ready.record(side)
        L.pending = (s0, s1, compare)

    def _await_groups(self, L):
        """Host side of _enqueue_groups: wait for the lane's read-back -> (groups per scenario and size class [s1-s0, classes]
        (copy), changed)."""
        s0, s1, compare = L.pending
        L.ev_ready.synchronize()
        return L.host_hist.numpy()[:s1 - s0].copy(), bool(compare and int(L.host_flag[0]) != 0)

    def _snapshot_groups(self, L, s0, s1):
        side, sl = self._side(L), slice(s0, s1)
        with torch.cuda.stream(side):
            self.ref_n[sl].copy_(self.n_groups[sl])
            self.ref_focal[sl].copy_(self.grp_focal[sl])
            self.ref_ids[sl].copy_(self.grp_ids[sl])

    def _enqueue_sim(self, L, t, s0, s1):
        """Simulator step of [s0, s1) on the lane's side stream, behind the sampled actions of the main stream."""
        side = self._side(L)
        if L.side is not None:
            L.ev_fwd.record(self._main)
            side.wait_event(L.ev_fwd)
        self.sim_step(t, s0=s0, s1=s1, stream=side.cuda_stream)
        if L.side is not None:
            L.ev_sim.record(side)
            L.sim_in_flight = True

    def _forward_waits(self, st):
        """The round-2 stream guard, OFF by default (forward_waits_for_sim): before a forward pass is queued on stream st, wait for the
        simulator steps of ALL lanes.  It kept a simulator step from running underneath another lane's matrix kernels while that
        overlap made rollouts irreproducible; the cause (SLP-vectorised code beside matrix-pipe workgroups) is
        gone from the build, and the wait would serialise the second pass on a side stream behind the other lane's first pass."""
        for L2 in self.lanes:
            if L2.sim_in_flight and self.forward_waits_for_sim:
                st.wait_event(L2.ev_sim)
                L2.sim_in_flight = False

    def _main_waits(self, L):
        if L.side is not None:
            self._main.wait_event(L.ev_ready)

    # ------------------------------------------------------------------ policy of one model batch
    def _ctx_index(self, L, s0, s1, st):
        lib, p = self.lib, _lib.ptr
        _lib.check(lib.ctrlsim_ctx_index_classes(s0, s1, self.N, self.dims.A, p(self.n_groups), p(self.grp_ids), p(self.own_g),
                                                 p(self.mem_g), len(self.sizes), self._sizes_c, p(L.ctx_scn), p(L.ctx_grp),
                                                 p(L.ctx_row0), p(self.ctx_of_group), p(self.own_ctx), p(self.own_slot),
                                                 p(self.mem_ctx), p(self.mem_slot), st), "ctx_index_classes")

    def _build_contexts(self, L, plan, t, Tq, tt_first, st):
        lib, p, d = self.lib, _lib.ptr, self.dims
        Tmax = self.steps
        if not plan:
            return
        n = len(plan)                                           # the plan's classes are back to back in the lane's context list
        assert plan[0][2] == 0 and all(plan[k + 1][2] == plan[k][2] + plan[k][0] for k in range(n - 1))
        Bs = (C.c_int * n)(*[q[0] for q in plan]); As = (C.c_int * n)(*[q[1] for q in plan])
        cs = (_lib.Ctx * n)(*[q[3] for q in plan])
        _lib.check(lib.ctrlsim_build_context_c(n, Bs, As, cs, self.N, d.T, t, Tq, tt_first, Tmax + 1, Tmax, self.P_all, d.P, d.NP,
                                               p(L.ctx_scn), p(L.ctx_grp), p(self.grp_focal), p(self.grp_ids),
                                               p(self.hist_states), p(self.hist_tok), p(self.hist_rtg), p(self.goals),
                                               p(self.types), p(self.roads), p(self.rtypes), self._zero4, st), "build_context")

    def _sample_rtg(self, L, t, s0, s1, st, noise_rtg=None):
        lib, p, d, sl = self.lib, _lib.ptr, self.dims, slice(s0, s1)
        _lib.check(lib.ctrlsim_sample_rtg_rows(p(L.rtg_logits), p(L.ctx_row0), d.R, p(self.own_ctx[sl]), p(self.own_slot[sl]),
                                               p(self.tilted[sl]), self.tilt,
                                               p(self.tilt_scn[sl]) if self.tilt_scn is not None else None,
                                               p(noise_rtg[sl]) if noise_rtg is not None else None, self.seed,
                                               p(self.scenario_id[sl]), t, p(self.hist_rtg[sl]), s1 - s0, self.N, self.steps, st),
                   "sample_rtg")

    def _sample_action(self, L, t, s0, s1, st, noise_act=None):
        lib, p, d, sl = self.lib, _lib.ptr, self.dims, slice(s0, s1)
        _lib.check(lib.ctrlsim_sample_action_rows(p(L.act_logits), p(L.ctx_row0), d.V, p(self.mem_ctx[sl]), p(self.mem_slot[sl]),
                                                  self.temperature, self.top_p,
                                                  p(noise_act[sl]) if noise_act is not None else None, self.seed,
                                                  p(self.scenario_id[sl]), t, p(self.hist_tok[sl]), p(self.act_now[sl]),
                                                  s1 - s0, self.N, self.steps, ZERO_ACTION_TOKEN, st), "sample_action")

    def _chunk_step_cached(self, L, s0, s1, counts, t):
        """Cached phase (t < T): policy step of one model batch against the decoder K/V cache it keeps in the lane's workspace."""
        lib, p, st, d = self.lib, _lib.ptr, self._main.cuda_stream, self.dims
        N, Tmax = self.N, self.steps
        Tq, tt_first = t + 1, max(t - 1, 0)
        plan, n, Bs, As, cs = self._class_plan(L, counts, d.T, Tq - tt_first)
        if L.side is not None and self.pass2_on_side and self.cached_on_side:
            # every kernel of a cached step touches a few rows per context: the whole step runs on the lane's OWN stream, so the
            # lanes' cached phases run side by side instead of taking turns on the main stream (which waits for the lane at
            # the first sliding step, _main_waits)
            st = L.side.cuda_stream
        self._ctx_index(L, s0, s1, st)
        self._build_contexts(L, plan, t, Tq, tt_first, st)
        self._forward_waits(L.side if st != self._main.cuda_stream else self._main)
        if n:
            _lib.check(lib.ctrlsim_dt_forward_pass1_cached_c(self.model.handle, n, Bs, As, cs, t, p(L.ws), p(L.rtg_logits), st),
                       "pass1_cached")
        self._sample_rtg(L, t, s0, s1, st)
        if n:
            _lib.check(lib.ctrlsim_dt_forward_pass2_c(self.model.handle, n, Bs, As, cs, Tq, t, N, Tmax, p(L.ctx_scn),
                                                      p(self.hist_rtg), p(L.ws), p(L.act_logits), 1, st), "pass2_cached")
        self._sample_action(L, t, s0, s1, st)

    def _dt_cfg(self):
        w, rc = self.w, self.cfg.nocturne.rew_cfg
        c = _lib.DtRewardCfg()
        c.pos_tol = float(rc["position_target_tolerance"])
        c.shaped_unit = float(rc.get("shaped_goal_distance_scaling", 1.0)) / float(rc["reward_scaling"])
        c.goal_mult, c.shaped_min, c.shaped_max = float(w.pos_target_achieved_rew_multiplier), float(w.pos_goal_shaped_min), float(w.pos_goal_shaped_max)
        c.veh_mult, c.max_veh_dist = float(w.veh_veh_collision_rew_multiplier), float(w.max_veh_veh_distance)
        c.edge_mult, c.edge_scale = float(w.veh_edge_collision_rew_multiplier), float(w.dist_to_road_edge_scaling_factor)
        for k, (lo, hi) in enumerate(((w.min_rtg_pos, w.max_rtg_pos), (w.min_rtg_veh, w.max_rtg_veh), (w.min_rtg_road, w.max_rtg_road))):
            c.rtg_lo[k], c.rtg_hi[k] = float(lo), float(hi)
        c.remove_shaped_goal, c.remove_shaped_veh, c.remove_shaped_edge = int(bool(w.remove_shaped_goal)), \
            int(bool(w.remove_shaped_veh_reward)), int(bool(w.remove_shaped_edge_reward))
        return c

    def _dt_ledger(self, t, s0, s1, st):
        """Decision-Transformer policy (cfgs/policy/dt.yaml): the RTG row of step t from the real-time rewards, on the device
        (ctrlsim_dt_ledger_step) — what PolicyEvaluator's per-vehicle bookkeeping feeds on the plugin surface."""
        p, sl = _lib.ptr, slice(s0, s1)
        if getattr(self, "_dt_cfg_c", None) is None:
            self._dt_cfg_c = self._dt_cfg()
        _lib.check(self.lib.ctrlsim_dt_ledger_step(s1 - s0, self.N, self.E, t, self.steps + 1, self.steps, p(self.hist_states[sl]),
                                                   p(self.coll[sl]), p(self.goals[sl]), p(self.edges[sl]),
                                                   p(self.dt_init_rtg[sl]) if self.dt_init_rtg is not None else None,
                                                   C.byref(self._dt_cfg_c), p(self.dt_ledger[sl]), p(self.dt_rtg_raw[sl]),
                                                   p(self.hist_rtg[sl]), st), "dt_ledger_step")

    def _policy_chunks(self, L, t, hist, lo, hi, noise_rtg=None, noise_act=None):
        """Full-recompute policy for scenarios [lo, hi) (hist = their group counts per size class), cut into model batches, on
        the main stream with lane L's buffers."""
        lib, p, st, d = self.lib, _lib.ptr, self._main.cuda_stream, self.dims
        N, Tmax = self.N, self.steps
        Tq = min(t, d.T - 1) + 1
        if d.VARIANT == 3 and self.device_ledger:
            self._dt_ledger(t, lo, hi, st)
        # The second pass touches Areg rows per context: a string of few-row kernels that leave most of the chip idle.  With a
        # side stream it runs THERE (sampling, second pass, sampling, then the simulator step that follows anyway), underneath the
        # other lane's first pass on the main stream; the lane's buffers are handed back to the main stream by an event.
        on_side = L.side is not None and not d.VARIANT and self.pass2_on_side
        first = True
        for (s0, s1, counts) in self._chunks(hist, lo):
            if on_side and not first:
                self._main.wait_event(L.ev_p2)               # the previous batch's second pass still reads the lane's buffers
            first = False
            plan, n, Bs, As, cs = self._class_plan(L, counts, Tq, Tq)
            self.full_pass_contexts += np.asarray(counts, np.int64)      # contexts per size class of the full-recompute passes (bench.py)
            if on_side and self.ctx_on_side:                     # (needs the second pass on the side stream too: stream order then keeps
                                                                 #  the next batch's context build behind this batch's readers of L.ctx)
                # the context index lists and tensors (two latency-bound kernels of 50-150 workgroups with long float64 chains: 2.6 % of
                # the step when they hold the main stream) are built on the lane's side stream, underneath the other lane's matrix
                # kernels; the forward pass waits for them by an event
                cst = L.side.cuda_stream
                self._ctx_index(L, s0, s1, cst)
                self._build_contexts(L, plan, t, Tq, 0, cst)
                L.ev_ctx.record(L.side)
                self._main.wait_event(L.ev_ctx)
            else:
                self._ctx_index(L, s0, s1, st)
                self._build_contexts(L, plan, t, Tq, 0, st)
            self._forward_waits(self._main)
            if n and d.VARIANT:                              # IL / Trajeglish: no RTG tokens, one forward (predict_rtgs False)
                _lib.check(lib.ctrlsim_dt_forward_actions(self.model.handle, plan[0][0], Tq, cs, p(L.ws), p(L.act_logits), st),
                           "forward_actions")
            elif n and on_side and self.tail_on_side:        # the few-row tail of the first pass goes to the side stream too
                _lib.check(lib.ctrlsim_dt_forward_pass1_c2(self.model.handle, n, Bs, As, cs, Tq, p(L.ws), p(L.rtg_logits), st,
                                                           L.side.cuda_stream), "pass1")
            elif n:
                _lib.check(lib.ctrlsim_dt_forward_pass1_c(self.model.handle, n, Bs, As, cs, Tq, p(L.ws), p(L.rtg_logits), None, st),
                           "pass1")
            st2 = st
            if on_side:
                L.ev_p1.record(self._main)                   # (also covers n == 0: sampling follows the context index kernels)
                L.side.wait_event(L.ev_p1)
                st2 = L.side.cuda_stream
            if not d.VARIANT:
                self._sample_rtg(L, t, s0, s1, st2, noise_rtg)
                if n:
                    _lib.check(lib.ctrlsim_dt_forward_pass2_c(self.model.handle, n, Bs, As, cs, Tq, t, N, Tmax, p(L.ctx_scn),
                                                              p(self.hist_rtg), p(L.ws), p(L.act_logits), 0, st2), "pass2")
            self._sample_action(L, t, s0, s1, st2, noise_act)
            if on_side:
                L.ev_p2.record(L.side)

    # ------------------------------------------------------------------ a lane's rollout as a generator
    def _lane_gen(self, L, lo, hi, steps, lane_idx=None):
        """Closed-loop rollout of scenarios [lo, hi) on lane L.  Yields wherever the host needs the lane's group counts
        back: the scheduler (run) then queues another lane's step before this one blocks on its read-back."""
        T = self.dims.T
        t0 = 0
        # pipelined jobs: the host never BLOCKS on a lane's read-back — a lane whose group counts are not back yet (its cached steps crawl
        # underneath the other lanes' full-row kernels) yields, so that the lanes in full-recompute steps keep the main stream fed
        poll = lane_idx is not None
        if self.use_cache and steps > 0:
            # steps 0 .. nT-1 chunk by chunk: while t < T the window starts at step 0, so a context's frame, membership and map
            # are constant and its decoder K/V can be cached across steps (csrc/forward.hip).  A chunk whose context set does
            # change (a vehicle stops existing) falls back to the full recompute.
            nT = min(T, steps)
            self._enqueue_groups(L, 0, lo, hi)
            yield
            while poll and not L.ev_ready.query():
                yield
            hist, _ = self._await_groups(L)
            self.groups_per_step[0, lo:hi] = hist.sum(1)
            for (s0, s1, counts) in self._chunks(hist, lo):
                cached_ok = sum(counts) > 0
                h = hist[s0 - lo:s1 - lo]
                if nT > 1:
                    self._snapshot_groups(L, s0, s1)
                for t in range(nT):
                    if lane_idx is not None:
                        self._lane_t[lane_idx] = t
                    if t > 0:
                        yield
                        while poll and not L.ev_ready.query():
                            yield
                        h, changed = self._await_groups(L)
                        self.groups_per_step[t, s0:s1] = h.sum(1)
                        cached_ok = cached_ok and not changed
                    self._main_waits(L)
                    if cached_ok:
                        self._chunk_step_cached(L, s0, s1, counts, t)
                    else:
                        self._policy_chunks(L, t, h, s0, s1)
                    self._enqueue_sim(L, t, s0, s1)
                    if t + 1 < nT:
                        self._enqueue_groups(L, t + 1, s0, s1, compare=True)
            t0 = nT
        if lane_idx is not None and self._max_sliding:
            # pipelined jobs (run_jobs): at most _max_sliding lanes run full-recompute steps at a time — a lane that has finished its cached
            # steps early (they run underneath the others' full-row kernels) parks here until a slot comes free
            while self._sliding >= self._max_sliding:
                yield
            self._sliding += 1
            self._holds_slot[lane_idx] = True
        if self.record_phases:                       # end of the lane's K/V-cached phase on its streams (bench.py: config.phases)
            for stream in (self._main, L.side):      # the cached steps' kernels run on the lane's side stream (cached_on_side) or on main
                if stream is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(stream)
                    self.phase_events[-1][1].append(ev)
        for t in range(t0, steps):
            if lane_idx is not None:
                self._lane_t[lane_idx] = t
            self._enqueue_groups(L, t, lo, hi)
            yield
            while poll and not L.ev_ready.query():
                yield
            hist, _ = self._await_groups(L)
            self.groups_per_step[t, lo:hi] = hist.sum(1)
            self._main_waits(L)
            self._policy_chunks(L, t, hist, lo, hi)
            self._enqueue_sim(L, t, lo, hi)

    def run(self, steps=None, noise_fn=None, s0=0, s1=None):
        """Roll scenarios [s0, s1) (default: all loaded) `steps` steps from their current state (reset() first for a fresh
        rollout).  noise_fn(t) -> (noise_rtg, noise_act): explicit sampling noise, synchronous single-lane path."""
        steps = self.steps if steps is None else steps
        s1 = self.S if s1 is None else s1
        self._bind()
        # what check_finite may repeat with the range-safe split: runs that started from reset(); anything else makes it raise
        self._unchecked.append((steps, s0, s1) if (noise_fn is None and getattr(self, "_fresh", None) == (s0, s1)) else None)
        if len(self._unchecked) > 4096:
            self._unchecked = [None]            # nobody checked for thousands of runs: not repeatable any more, still loud
        self._fresh = None
        if noise_fn is not None:
            assert s0 == 0 and s1 == self.S
            for t in range(steps):
                nr, na = noise_fn(t)
                self.step(t, nr, na)
            return self
        main = self._main = torch.cuda.current_stream(self.device)
        if self.record_phases:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(main)
            self.phase_events.append([ev0, [], None])
        n = min(self.n_lanes, max(1, s1 - s0))
        cuts = [s0 + (s1 - s0) * i // n for i in range(n + 1)]
        gens = []
        for L, lo, hi in zip(self.lanes, cuts[:-1], cuts[1:]):
            if L.side is not None:
                L.side.wait_stream(main)
            gens.append(self._lane_gen(L, lo, hi, steps))
        while gens:                                   # round robin: each lane runs to its next read-back point
            for g in list(gens):
                try:
                    next(g)
                except StopIteration:
                    gens.remove(g)
        for L in self.lanes[:n]:
            if L.side is not None:
                main.wait_stream(L.side)
        if self.record_phases:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(main)
            self.phase_events[-1][2] = ev1
        return self

    def run_jobs(self, jobs, steps=None, stagger=True, max_sliding=None):
        """Pipelined rollouts (round 4): every job (s0, s1) — a scenario range — is reset and rolled `steps` steps by ONE lane, and the
        lanes take jobs from the list as they come free.  With `stagger`, lane i > 0 starts its first job only when lane 0 has left the
        K/V-cached steps of ITS first job (t = T): from then on the lanes stay about half a rollout apart, so that one lane's cached
        phase — strings of few-row kernels on its side stream that leave most of the chip idle — always runs underneath another lane's
        full-recompute steps instead of beside the other lane's cached phase (run() starts every lane at t = 0 together:
        policies/autoregressive_policy.py:55-70 is the window rule that creates the two phases).  Scenario results do not depend on
        which lane rolls them or on what runs beside them (tests/test_gpu_sim_ctx.py)."""
        steps = self.steps if steps is None else steps
        jobs = [(int(a), int(b)) for a, b in jobs if b > a]
        self._bind()
        self._fresh = None
        main = self._main = torch.cuda.current_stream(self.device)
        for a, b in jobs:
            self._unchecked.append((steps, a, b))
        if len(self._unchecked) > 4096:
            self._unchecked = [None]
        lanes = self.lanes[:max(1, min(self.n_lanes, len(jobs)))]
        todo = list(reversed(jobs))
        self._lane_t = [0] * len(self.lanes)
        nT = min(self.dims.T, steps) if self.use_cache else 0
        done0, first0 = [False], [False]
        # max_sliding (default: 2 when there are more lanes than that): the EXTRA lanes run the cached steps of the next jobs ahead of time
        self._max_sliding = int(max_sliding) if max_sliding else (2 if len(lanes) > 2 else 0)
        self._sliding, self._holds_slot = 0, [False] * len(self.lanes)
        if self._max_sliding:
            stagger = False                                  # the slots stagger the lanes by themselves
        if steps <= nT:
            stagger = False                                  # a rollout that never leaves the cached steps has no second phase to stagger against
        # run()'s phase records describe ONE range rolled by all lanes together; here every lane has its own jobs: one record per call,
        # the lanes' "left the cached phase" events all land in it (phase_times() takes the last one)
        rec = None
        if self.record_phases:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(main)
            rec = [ev0, [], None]
            self.phase_events.append(rec)

        def lane_loop(L, idx):
            if L.side is not None:
                L.side.wait_stream(main)
            if stagger and idx > 0:
                # lane 0 still inside the cached steps of its FIRST job (released as well when that job is over — _lane_t is reset
                # per job — or when lane 0 has nothing left to do)
                while not done0[0] and not first0[0] and self._lane_t[0] < nT:
                    yield
            while todo:
                lo, hi = todo.pop()
                with torch.cuda.stream(self._side(L)):             # the job's reset in the lane's own stream order
                    self.reset(lo, hi)
                self._fresh = None
                self._lane_t[idx] = 0
                yield from self._lane_gen(L, lo, hi, steps, idx)
                if idx == 0:
                    first0[0] = True
                if self._holds_slot[idx]:
                    self._holds_slot[idx] = False
                    self._sliding -= 1
            if idx == 0:
                done0[0] = True

        gens = [lane_loop(L, i) for i, L in enumerate(lanes)]
        while gens:
            for g in list(gens):
                try:
                    next(g)
                except StopIteration:
                    gens.remove(g)
        for L in lanes:
            if L.side is not None:
                main.wait_stream(L.side)
        if rec is not None:
            rec[2] = torch.cuda.Event(enable_timing=True)
            rec[2].record(main)
        self._max_sliding = 0
        return self

    def phase_times(self):
        """(seconds until the LAST lane left its K/V-cached phase, seconds after that) summed over the recorded runs, in main-stream
        time (call after a synchronise; record_phases must have been set before the runs)."""
        cached = sliding = 0.0
        for ev0, mids, ev1 in self.phase_events:
            if ev1 is None:
                continue
            mid = max((ev0.elapsed_time(m) for m in mids), default=0.0)
            tot = ev0.elapsed_time(ev1)
            cached += mid * 1e-3
            sliding += (tot - mid) * 1e-3
        return cached, sliding

    # ------------------------------------------------------------------ synchronous single-stream steps (plugin surface)
    def step(self, t, noise_rtg=None, noise_act=None):
        """One closed-loop step: policy (grouping, contexts, two-pass model, sampling) then the simulator step.
        noise_rtg [S,N,3,R] / noise_act [S,N,V] float32 tensors (explicit Exp(1) noise) or None (in-kernel)."""
        self.policy_step(t, noise_rtg, noise_act)
        self.sim_step(t)

    def policy_step(self, t, noise_rtg=None, noise_act=None):
        """AutoregressivePolicy.predict for every scenario: writes hist_rtg[..., t, :], hist_tok[..., t], act_now."""
        self._main = torch.cuda.current_stream(self.device)
        self._bind()
        L = self.lanes[0]
        side, L.side = L.side, None                   # everything on the caller's stream
        try:
            self._enqueue_groups(L, t, 0, self.S)
            hist, _ = self._await_groups(L)
        finally:
            L.side = side
        self.groups_per_step[t] = hist.sum(1)
        self._policy_chunks(L, t, hist, 0, self.S, noise_rtg, noise_act)

    def metrics_pack(self, gt, goals4, eval_mask=None, out=None):
        """Evaluator statistics of the loaded scenarios' finished rollouts, accumulated on the device (ctrlsim_metrics_pack):
        -> float64 device tensor in MetricAccumulators.pack() order (add into `out` if given).  gt [S,N,T1,5] float64 = logged
        x, y, heading, speed, exist; goals4 [S,N,4] float64 = goal x, y, heading, speed; eval_mask [S,N] uint8 or None."""
        from .metrics import MetricAccumulators
        dev, w, p = self.device, self.w, _lib.ptr
        n = int(self.lib.ctrlsim_metrics_size())
        if out is None:
            out = torch.zeros(n, dtype=torch.float64, device=dev)
        E = MetricAccumulators.EDGES
        edges = torch.from_numpy(np.concatenate([E["lin"], E["ang"], E["accel"], E["nd"]]).astype(np.float64)).to(dev)
        as_dev = lambda a, dt: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, dt).contiguous()
        gt_d, g4_d = as_dev(gt, torch.float64), as_dev(goals4, torch.float64)
        em_d = as_dev(eval_mask, torch.uint8) if eval_mask is not None else None
        assert tuple(gt_d.shape) == (self.S, self.N, self.steps + 1, 5) and tuple(g4_d.shape) == (self.S, self.N, 4)
        params = (C.c_double * 5)(float(self.cfg.nocturne.rew_cfg["position_target_tolerance"]), w.min_accel, w.max_accel,
                                  w.accel_discretization, w.steer_discretization)
        _lib.check(self.lib.ctrlsim_metrics_pack(self.S, self.N, self.steps + 1, self.steps, int(self.cfg.nocturne.history_steps),
                                                 float(self.dt), p(self.hist_states), p(self.coll), p(self.hist_tok), p(gt_d),
                                                 p(g4_d), p(em_d), params, p(edges), p(out), _lib.stream_ptr()), "metrics_pack")
        return out

    def check_finite(self):
        """Synchronise and look at this engine's guard counter.  With split="auto", runs that started from reset() and met
        non-finite values under the two-fp16-plane split are ALL repeated (every range rolled since the last check, not only the
        last one) with three bf16 planes, which stay selected for this engine and later engines of the model; -> True if that
        happened.  Anything else raises, naming the ranges."""
        bad = self.nonfinite()
        runs, self._unchecked = self._unchecked, []
        if not bad:
            return False
        if bad >= 65536:
            # simulator events (the second word of the guard pair, reported in the high half: csrc/common.h): contacts beyond the island solver's table.  Not a matter
            # of the operand split — no fallback, no permanent switch of the model to three planes: raise at once
            raise FloatingPointError(f"{bad >> 16} simulator contacts beyond the island solver's table (csrc/sim.hip: MAX_ISLAND_CONTACTS) in the "
                                     f"rollouts of scenario ranges {[r[1:] if r else '?' for r in runs]}"
                                     + (f"; also {bad & 65535} non-finite events" if bad & 65535 else ""))
        if self.split == "auto" and self.scheme == 1 and runs and all(r is not None for r in runs):
            self._set_split(0)
            for steps, s0, s1 in runs:
                self.reset(s0, s1)
                self.run(steps, s0=s0, s1=s1)
            bad = self.nonfinite()
            self._unchecked = []
            if not bad:
                return True
        raise FloatingPointError(f"{bad} guard events in the rollouts of scenario ranges {[r[1:] if r else '?' for r in runs]}: sampling "
                                 "races without a finite logit / LayerNorm rows with a non-finite variance (an activation beyond the "
                                 "fp16 range of the split operands, csrc/split.h, or bad weights), or simulator contacts beyond the "
                                 "island solver's table (csrc/sim.hip: MAX_ISLAND_CONTACTS)")

    def rollout(self, steps=None, s0=0, s1=None):
        """reset + run + check_finite of scenarios [s0, s1): a complete closed-loop rollout, repeated with the range-safe operand
        split if the fast one overflowed (split="auto")."""
        self.reset(s0, s1)
        self.run(steps, s0=s0, s1=s1)
        self.check_finite()
        return self

    def results(self):
        self.check_finite()
        return dict(tokens=self.hist_tok.cpu().numpy(), rtg_bins=self.hist_rtg.cpu().numpy(),
                    states=self.hist_states.cpu().numpy(), coll=self.coll.cpu().numpy(),
                    n_groups=self.groups_per_step.copy())
