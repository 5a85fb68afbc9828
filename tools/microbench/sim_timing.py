"""Phase breakdown of sim_step_kernel (build sim.hip with -DSIM_TIMING: CTRLSIM_EXTRA_DEFS=-DSIM_TIMING python ctrl-sim_amd/csrc/build.py --force).
Rolls S scenarios of the bench shape with the policy and prints the share of block time per phase (s_memtime on thread 0)."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import ctrlsim_amd  # noqa
from ctrlsim_amd import spec, weights, scenarios, _lib
from ctrlsim_amd.engine import RolloutEngine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = spec.make_cfg(); d = spec.Dims(cfg)
eng = RolloutEngine(cfg, weights.generate(d, 0), 'cuda:0', max_ctx=512, seed=0)
eng.load_scenarios(scenarios.make_batch(0, range(S), n_agents=64, n_polylines=512), steps=steps)
lib = _lib.lib()
out = (C.c_ulonglong * 12)()
lib.ctrlsim_sim_timing(out, 1)
eng.run(steps); torch.cuda.synchronize()
lib.ctrlsim_sim_timing(out, 0)
v = np.array(list(out), float)
names = ["freecar+tree load", "teleports", "contact update", "single-body islands", "contact islands (lane 0)", "sync fixtures",
         "tree moves (lane 0)", "find_new_contacts (lane 0)", "write-back", "collision: edge tests + flag stores",
         "collision: corners + history row", "collision: vehicle pairs"]
for n, x in zip(names, v):
    print(f"sim {n:32s} {100 * x / v.sum():6.2f} %   {x / (S * steps) / 100:8.1f} us per block-step (100 MHz clock)")
