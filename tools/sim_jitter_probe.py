"""Intra-workgroup race probe of sim_step: with a library built with -DSIM_JITTER (random per-wave stalls around every workgroup
barrier of the step) single-stream rollouts must still be bit-identical from run to run.   python tools/sim_jitter_probe.py [runs=8]"""
import sys

sys.path.insert(0, '.')
import numpy as np

import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import spec, weights, scenarios
from ctrlsim_amd.engine import RolloutEngine

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = spec.make_cfg(nocturne__steps=90, nocturne__history_steps=1)
d = spec.Dims(cfg)
w = weights.generate(d, 0)
scns = [scenarios.make_scenario(7, i, n_agents=64, n_polylines=512) for i in range(3)]
model, ref, bad = None, None, 0
for k in range(runs + 1):
    eng = RolloutEngine(cfg, w, 'cuda:0', max_ctx=64, seed=3, model=model, lanes=1)
    model = eng.model
    eng.load_scenarios(scns, steps=90)
    r = eng.run(90).results()
    cur = (r["tokens"].copy(), r["states"].copy(), r["coll"].copy())
    if ref is None:
        ref = cur
        continue
    nt, ds = int((ref[0] != cur[0]).sum()), float(np.abs(ref[1] - cur[1]).max())
    bad += nt > 0 or ds > 0
    print(f"run {k}: token differences {nt}, max |state difference| {ds}, flag differences {int((ref[2] != cur[2]).sum())}", flush=True)
print(f"{bad} of {runs} single-stream runs differ from the first")
