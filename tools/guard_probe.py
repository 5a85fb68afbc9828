"""Out-of-bounds READ probe of the engine's state arrays: every device tensor the engine owns (simulator state, histories, group
tables, ...) is re-homed into the middle of a slab whose margins (64 KB each side) hold a byte pattern; single-stream rollouts
with different patterns must be bit-identical — a kernel that reads past an array's end (or before its start) would pick the
pattern up.  (A stray WRITE shows as a changed margin.)    python tools/guard_probe.py"""
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import spec, weights, scenarios
from ctrlsim_amd.engine import RolloutEngine

G = 1 << 16
cfg = spec.make_cfg(nocturne__steps=90, nocturne__history_steps=1)
d = spec.Dims(cfg)
w = weights.generate(d, 0)
scns = [scenarios.make_scenario(7, i, n_agents=64, n_polylines=512) for i in range(3)]
model = None


def rehome(eng, byte):
    slabs = []
    for k, v in list(eng.__dict__.items()):
        if isinstance(v, torch.Tensor) and v.is_cuda and v.is_contiguous() and v.numel() > 0:
            nb = v.numel() * v.element_size()
            pad = (-nb) % 256
            slab = torch.full((G + nb + pad + G,), byte, dtype=torch.uint8, device=v.device)
            view = slab[G:G + nb].view(v.dtype).view(v.shape)
            view.copy_(v)
            setattr(eng, k, view)
            slabs.append((k, slab, nb))
    return slabs


def run(byte):
    global model
    eng = RolloutEngine(cfg, w, 'cuda:0', max_ctx=64, seed=3, model=model, lanes=1)
    model = eng.model
    eng.load_scenarios(scns, steps=90)
    slabs = rehome(eng, byte) if byte is not None else []
    r = eng.run(90).results()
    torch.cuda.synchronize()
    for k, slab, nb in slabs:
        m = torch.cat([slab[:G], slab[G + nb + ((-nb) % 256):]])
        bad = int((m != byte).sum())
        if bad:
            print(f"  STRAY WRITE next to {k}: {bad} margin bytes changed")
    return r["tokens"].copy(), r["states"].copy()


ref = run(None)
for byte in (0x00, 0xFF, 0x7F, 0x3F):
    b = run(byte)
    print(f"margin byte {byte:#04x}: token differences {int((ref[0] != b[0]).sum())}, max |state difference| "
          f"{float(np.abs(ref[1] - b[1]).max())}", flush=True)
