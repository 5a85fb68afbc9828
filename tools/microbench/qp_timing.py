"""Sustained time of ctrlsim_outproj_ln_q (the QP mode of the fused feed-forward kernel) at a full-row and a rollout-sized launch, with its
error against float64 — for A/B runs of build variants (CTRLSIM_LIB).  usage: python tools/microbench/qp_timing.py"""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import torch
import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib
from ctrlsim_amd.pack import outproj_q_planes

DEV = 'cuda:0'
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()


def sustained(fn, secs=0.8):
    def burst():
        n, t0 = 0, time.perf_counter()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        while True:
            for _ in range(8):
                fn()
            n += 8
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > secs:
                break
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    burst()
    return burst()


g = torch.Generator().manual_seed(1)
Wo = torch.randn(256, 256, generator=g) * 0.07; Wq = torch.randn(256, 256, generator=g) * 0.09
bo, g0, be0, bq = (torch.randn(256, generator=g).to(DEV) * s for s in (0.3, 1.0, 0.2, 0.4))
dev = lambda a: torch.from_numpy(a.view(np.int16).copy()).to(DEV)
wod, wqd = (dev(a) for a in outproj_q_planes(Wo.numpy(), Wq.numpy(), 1))
out = []
for M in (256 * 2304, 48 * 2304):
    O = torch.randn(M, 256, generator=g).to(DEV); R = torch.randn(M, 256, generator=g).to(DEV)
    X1 = torch.empty_like(R); Q = torch.empty_like(R)
    call = lambda: lib.ctrlsim_outproj_ln_q(p(O), 256, p(R), 256, p(wod), p(bo), p(g0), p(be0), p(wqd), p(bq), p(X1), 256, p(Q), 256, M, st)
    call(); torch.cuda.synchronize()
    n = 4096
    x1 = torch.nn.functional.layer_norm(R[:n].double() + O[:n].double() @ Wo.to(DEV).double().T + bo.double(), (256,), g0.double(), be0.double(), 1e-5)
    q = x1 @ Wq.to(DEV).double().T + bq.double()
    e1 = (X1[:n].double() - x1).abs().max().item(); e2 = (Q[:n].double() - q).abs().max().item()
    out.append(f"M={M}: {sustained(call):.3f} ms (err x1 {e1:.1e} q {e2:.1e})")
print(" | ".join(out))
