"""`CtRLSim` — the model object of the plugin surface (reference: models/ctrl_sim.py:19-45).

The reference class is a LightningModule whose only rollout-relevant members are `.cfg`, `.eval()` and
`forward(data, eval) -> {'action_preds','rtg_preds','state_preds'}`; `load_from_checkpoint(path)` builds it from a
Lightning checkpoint whose `hyper_parameters` carry the cfg (eval_sim.py:52, policies/policy.py:28-29).
Here the object owns the packed device weights (`HipModel`); the HIP forward is driven by the policy/engine through
the C ABI; `forward` on reference-layout tensors is the reference's return contract ([B,A,T,.] logits of every head,
teacher-forced), or — with `token_index` — the two-pass logits of one timestep, the slice AutoregressivePolicy reads."""
from __future__ import annotations

import numpy as np

from ..spec import Dims, check_supported
from .. import weights as _weights


class CtRLSim:
    def __init__(self, cfg, weights=None, seed=0, device="cuda:0"):
        self.cfg = cfg
        check_supported(cfg)                    # e.g. a checkpoint trained with use_map=False: another network
        self.dims = Dims(cfg)
        self.weights = weights if weights is not None else _weights.generate(self.dims, seed)
        self.device = device
        self._hip = None
        self.training = False

    @classmethod
    def load_from_checkpoint(cls, path, cfg=None, device="cuda:0"):
        """Lightning checkpoint: {'state_dict': {...}, 'hyper_parameters': {'cfg': ...}} (models/ctrl_sim.py:23-25)."""
        import torch
        ck = torch.load(path, map_location="cpu", weights_only=False)
        if cfg is None:
            cfg = ck["hyper_parameters"]["cfg"]
        check_supported(cfg)
        d = Dims(cfg)
        return cls(cfg, _weights.from_state_dict(d, ck["state_dict"]), device=device)

    def eval(self):
        self.training = False
        return self

    def state_dict(self):
        return dict(self.weights)

    @property
    def hip(self):
        if self._hip is None:
            from ..engine import HipModel
            self._hip = HipModel(self.cfg, self.weights, self.device)
        return self._hip

    def __call__(self, data, eval=False, token_index=None):
        return self.forward(data, eval, token_index)

    def _arrays(self, data):
        ag, mp = data["agent"], data["map"]
        g = lambda o, k: (o[k] if isinstance(o, dict) else getattr(o, k))
        host = lambda v: np.asarray(v.cpu() if hasattr(v, "cpu") else v)
        arrs = {k: host(g(ag, k)) for k in ("agent_states", "agent_types", "goals", "actions", "rtgs", "timesteps")}
        arrs["road_points"], arrs["road_types"] = host(g(mp, "road_points")), host(g(mp, "road_types"))
        return arrs

    def forward(self, data, eval=False, token_index=None):
        """data: reference MotionData-like mapping (data['agent'].agent_states ...).
        token_index None (the reference's contract, models/ctrl_sim.py:41-45 + decoder.py:52-77): one teacher-forced forward,
        {'action_preds': [B,A,T,V], 'rtg_preds': [B,A,T,R*C], 'state_preds': [B,A,T,2T]} (the keys the model's heads provide).
        token_index given: the logits of that window step for every slot, {'rtg_preds': [B,A,R*C], 'action_preds': [B,A,V]},
        from the two-pass HIP forward with the rtg bins already present in data (what the reference's second call computes)."""
        import ctypes as C
        import torch
        from .. import _lib
        from ..engine import ctx_from_reference_layout
        d = self.dims
        arrs = self._arrays(data)
        B = arrs["agent_states"].shape[0]
        dev = self.device
        lib, st = _lib.lib(), _lib.stream_ptr()
        if token_index is None:
            cb = ctx_from_reference_layout(d, arrs, d.T, dev)
            cb.slot_gid.copy_(torch.arange(d.A, dtype=torch.int32, device=dev).expand(B, d.A))
            ws = torch.empty(self.hip.workspace_bytes(B, d.T), dtype=torch.uint8, device=dev)
            act = torch.empty(B, d.T, d.A, d.V, device=dev)
            rtg = torch.empty(B, d.T, d.A, d.R * d.C, device=dev) if d.VARIANT == 0 else None
            fut = torch.empty(B, d.T, d.A, d.FUT, device=dev) if "decoder.predict_future_states.mlp.0.weight" in self.weights else None
            _lib.check(lib.ctrlsim_forward_all(self.hip.handle, B, d.T, C.byref(cb.struct), ws.data_ptr(), act.data_ptr(),
                                               rtg.data_ptr() if rtg is not None else None,
                                               fut.data_ptr() if fut is not None else None, st), "forward_all")
            torch.cuda.synchronize()
            out = {"action_preds": act.permute(0, 2, 1, 3)}
            if fut is not None:
                out["state_preds"] = fut.permute(0, 2, 1, 3)
            if rtg is not None:
                out["rtg_preds"] = rtg.permute(0, 2, 1, 3)
            return out
        ti = token_index if token_index >= 0 else d.T + token_index
        Tq = ti + 1
        cb = ctx_from_reference_layout(d, arrs, Tq, dev)
        cb.slot_gid.copy_(torch.arange(d.A, dtype=torch.int32, device=dev).expand(B, d.A))
        ws = torch.empty(self.hip.workspace_bytes(B, Tq), dtype=torch.uint8, device=dev)
        rtg = torch.empty(B, d.A, d.R * d.C, device=dev)
        act = torch.empty(B, d.A, d.V, device=dev)
        hist = torch.zeros(B, d.A, 1, 3, dtype=torch.int32, device=dev)
        hist[:, :, 0] = torch.from_numpy(arrs["rtgs"][:, :, ti].astype(np.int32)).to(dev)
        scn = torch.arange(B, dtype=torch.int32, device=dev)
        _lib.check(lib.ctrlsim_dt_forward_pass1(self.hip.handle, B, Tq, C.byref(cb.struct), ws.data_ptr(), rtg.data_ptr(), None, st))
        _lib.check(lib.ctrlsim_dt_forward_pass2(self.hip.handle, B, Tq, 0, d.A, 1, C.byref(cb.struct), scn.data_ptr(),
                                                hist.data_ptr(), ws.data_ptr(), act.data_ptr(), 0, st))
        torch.cuda.synchronize()
        return {"rtg_preds": rtg, "action_preds": act}
