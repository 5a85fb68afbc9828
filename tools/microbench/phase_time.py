"""Wall time of the K/V-cached phase (steps 0..31) vs the sliding-window phase of one bench slice.  Usage: phase_time.py [S=102]"""
import sys, time
sys.path.insert(0, '.')
import torch
import ctrlsim_amd  # noqa
from ctrlsim_amd import spec, weights, scenarios
from ctrlsim_amd.engine import RolloutEngine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 102
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = spec.make_cfg(); d = spec.Dims(cfg)
eng = RolloutEngine(cfg, weights.generate(d, 0), 'cuda:0', max_ctx=512, seed=0, lanes=LANES)
eng.load_scenarios(scenarios.make_batch(0, range(S), n_agents=64, n_polylines=512), steps=90)
for steps in (32, 32, 90):
    eng.reset(0, S); torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.run(steps, s0=0, s1=S); torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"phase lanes={LANES} steps={steps:3d}: {t1 - t0:.3f} s", flush=True)
import cProfile, pstats
eng.reset(0, S); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); eng.run(32, s0=0, s1=S); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime")
import io; buf = io.StringIO(); st.stream = buf; st.print_stats(14); print("\n".join("phase " + l for l in buf.getvalue().splitlines() if l.strip())[:4000])
