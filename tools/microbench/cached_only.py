"""The K/V-cached phase alone (steps 0..31 of a slice), for rocprofv3 --kernel-trace --stats.  Usage: cached_only.py [S=102] [lanes=2] [reps=3]"""
import sys, time
sys.path.insert(0, '.')
import torch
import ctrlsim_amd  # noqa
from ctrlsim_amd import spec, weights, scenarios
from ctrlsim_amd.engine import RolloutEngine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 102
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = spec.make_cfg(); d = spec.Dims(cfg)
eng = RolloutEngine(cfg, weights.generate(d, 0), 'cuda:0', max_ctx=512, seed=0, lanes=LANES)
eng.load_scenarios(scenarios.make_batch(0, range(S), n_agents=64, n_polylines=512), steps=90)
for rep in range(REPS):
    eng.reset(0, S); torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run(32, s0=0, s1=S); torch.cuda.synchronize()
    print(f"cached phase lanes={LANES} rep {rep}: {time.perf_counter() - t0:.3f} s", flush=True)
