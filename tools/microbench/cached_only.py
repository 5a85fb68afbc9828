import sys
sys.path.insert(0, '.')
import torch
import ctrlsim_amd  # noqa
from ctrlsim_amd import spec, weights, scenarios
from ctrlsim_amd.engine import RolloutEngine
S = 102
cfg = spec.make_cfg(); d = spec.Dims(cfg)
eng = RolloutEngine(cfg, weights.generate(d, 0), 'cuda:0', max_ctx=512, seed=0, lanes=2)
eng.load_scenarios(scenarios.make_batch(0, range(S), n_agents=64, n_polylines=512), steps=90)
for _ in range(3):
    eng.reset(0, S); eng.run(32, s0=0, s1=S); torch.cuda.synchronize()
