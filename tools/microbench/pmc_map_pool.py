"""Three launches of map_pool_kernel at the rollout shape (B = 1024 contexts x 200 polylines, ragged point counts as in the synthetic scenes)
for the SQ counter passes of profiles/r05_c_pmc_map_pool.md."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np
import torch

import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import _lib, spec, weights
from ctrlsim_amd.engine import HipModel, CtxBuffers

DEV = 'cuda:0'
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = spec.make_cfg(); d = spec.Dims(cfg)
model = HipModel(cfg, weights.generate(d, 0), DEV)
lib, p, st = _lib.lib(), _lib.ptr, _lib.stream_ptr()
rs = np.random.RandomState(0)
npts = rs.randint(20, d.NP + 1, (B, d.P))
ex = (np.arange(d.NP)[None, None] < npts[..., None]).astype(np.float32)
rp = np.concatenate([rs.randn(B, d.P, d.NP, 2).astype(np.float32) * 20 * ex[..., None], ex[..., None]], -1)
cb = CtxBuffers(d, B, DEV)
cb.road_pts.copy_(torch.from_numpy(rp).to(DEV))
cb.road_types.zero_(); cb.road_types[..., 1] = 1
out = torch.empty(B * d.P, d.D, device=DEV); pad = torch.empty(B, d.P, dtype=torch.uint8, device=DEV)
for _ in range(3):
    _lib.check(lib.ctrlsim_map_pool(model.handle, B, p(cb.road_pts), p(out), p(pad), st))
torch.cuda.synchronize()
print("visible points per launch", int(ex.sum()), "of", B * d.P * d.NP)
