"""Parity sweep outside the test suite: N random scenes through the HIP engine and the CPU oracle, counts scenes with any\ndifferent sampled token / RTG bin.  usage (GPU box): python tools/flip_rate.py [n_scenes] [loop|full]   (about 11 s of oracle time per scene)"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np
from helpers import cfg_of
import rollout_oracle, sim_libs
from ctrlsim_amd import spec, scenarios, weights
from ctrlsim_amd.engine import RolloutEngine
kind = sys.argv[2] if len(sys.argv) > 2 else "loop"           # "full" = the real model size (slow oracle: use few scenes)
cfg = cfg_of(kind); d = spec.Dims(cfg); w = weights.generate(d, 0)
n_scn, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40, (12 if kind == "loop" else 8)
scns = [scenarios.make_scenario(77, i, n_agents=10, n_polylines=14 if kind == "loop" else 230, n_points=d.NP, extent=38.0)
        for i in range(n_scn)]
eng = RolloutEngine(cfg, w, "cuda:0", max_ctx=256, seed=11, tilt=(5.0, -10.0, 10.0))
eng.load_scenarios(scns, steps=steps)
r = eng.run(steps).results()
ro = rollout_oracle.RolloutOracle(cfg, w, seed=11, tilt=(5.0, -10.0, 10.0))
tot = bad_scn = 0; first = []
t0 = time.time()
for s, scn in enumerate(scns):
    o = ro.run(scn, steps, sim_libs.OracleSim)
    eq_tok = r["tokens"][s][:, :steps] == o["tokens"]; eq_rtg = (r["rtg_bins"][s][:, :steps] == o["rtg_bins"]).all(-1)
    ok = eq_tok & eq_rtg
    tot += ok.size
    if not ok.all():
        bad_scn += 1
        t_first = int(np.where(~ok.all(0))[0][0])
        first.append((s, t_first, int((~ok[:, t_first]).sum())))
print(f"{n_scn} scenes x {steps} steps x 10 vehicles: scenes with any token/RTG difference vs the oracle: {bad_scn}; first divergences (scene, step, vehicles): {first}; samples per scene {steps*10*4}; oracle time {time.time()-t0:.0f}s")
