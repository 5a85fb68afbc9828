"""map_pool A/B (round 6: packed-fp32 kernel vs the scalar one).  Runs ctrlsim_map_pool of the library selected with CTRLSIM_LIB on seeded
scenes (ragged: 20..NP visible points per polyline, some polylines empty; and full), dumps the output rows and times the launch.
Usage: CTRLSIM_LIB=<.so> python tools/microbench/map_ab.py <out.npz> [contexts=1024]"""
import sys
import time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np
import torch
import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib, spec, weights
from ctrlsim_amd.engine import HipModel, CtxBuffers

DEV = 'cuda:0'
out_path = sys.argv[1]
Bm = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
lib = _lib.lib(); p = _lib.ptr; st = _lib.stream_ptr()
cfg = spec.make_cfg(); d = spec.Dims(cfg)
model = HipModel(cfg, weights.generate(d, 0), DEV)
res = {}
for tag in ("ragged", "full"):
    rs = np.random.RandomState(0)
    npts = rs.randint(20, d.NP + 1, (Bm, d.P)) if tag == "ragged" else np.full((Bm, d.P), d.NP)
    if tag == "ragged":
        npts[rs.uniform(size=npts.shape) < 0.05] = 0                       # empty polylines: point 0 un-masked
    ex = (np.arange(d.NP)[None, None] < npts[..., None]).astype(np.float32)
    if tag == "ragged":                                                    # holes inside a polyline too
        ex *= (rs.uniform(size=ex.shape) > 0.1)
    rp = np.concatenate([rs.randn(Bm, d.P, d.NP, 2).astype(np.float32) * 20 * ex[..., None], ex[..., None]], -1)
    cb = CtxBuffers(d, Bm, DEV)
    cb.road_pts.copy_(torch.from_numpy(rp).to(DEV))
    out = torch.empty(Bm * d.P, d.D, device=DEV); padm = torch.empty(Bm, d.P, dtype=torch.uint8, device=DEV)
    fn = lambda: lib.ctrlsim_map_pool(model.handle, Bm, p(cb.road_pts), p(out), p(padm), st)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 20)
    print(f"map_pool {tag:7s} {Bm} contexts x {d.P} polylines x {d.NP} points: {best:.3f} ms", flush=True)
    k = min(Bm, 64)
    res[tag + "_out"] = out[:k * d.P].cpu().numpy(); res[tag + "_pad"] = padm[:k].cpu().numpy(); res[tag + "_ms"] = best
np.savez(out_path, **res)
