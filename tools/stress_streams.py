"""Reproducibility stress of the engine's stream options ("Few-row kernels underneath the full-row ones").

Rolls three full-size scenarios (64 vehicles x 512 polylines x 90 steps) once on a single stream and then `runs` times with two
lanes and the chosen side-stream switches; every run must reproduce the single-stream tokens and trajectories bit for bit.
    python tools/stress_streams.py [runs=12] [p2=0] [tail=0] [cached=0] [kernels=111]
With all switches 0 (the defaults) 40 of 40 runs were identical on MI355X; with p2=1 about one run in three was not.
sim_delay_us: before the engine's _forward_waits guard existed, 600-1500 us made 8 of 72 default-mode runs differ (0 of 72 with it)."""
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import spec, weights, scenarios
from ctrlsim_amd.engine import RolloutEngine

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
p2, tail, cached = (bool(int(sys.argv[i])) if len(sys.argv) > i else False for i in (2, 3, 4))
# kernel selection (ctrlsim_set_option 0 / 1 / 3): "a g f" digits, e.g. 101 = split attention, f32 GEMMs, fused FFN flag on; default 111
sel = sys.argv[5] if len(sys.argv) > 5 else "111"
pollute = len(sys.argv) > 6 and bool(int(sys.argv[6]))
guard = not (len(sys.argv) > 8 and sys.argv[8] == '0')        # 0: without the engine's _forward_waits guard
contacts = not (len(sys.argv) > 9 and sys.argv[9] == '0')     # 0: simulator without the Box2D contact path
delay_us = int(sys.argv[7]) if len(sys.argv) > 7 else 0      # the simulator step waits this long on its stream first (it then
                                                            # overlaps later kernels of the other lane's step)
f32 = sel[0] == "0"                                       # f32 attention has no compact contexts
cfg = spec.make_cfg(nocturne__steps=90, nocturne__history_steps=1)
d = spec.Dims(cfg)
w = weights.generate(d, 0)
import os
n_scn = int(os.environ.get("STRESS_SCENARIOS", "3"))      # more scenarios = more simulator workgroups per launch = more exposure per run
scns = [scenarios.make_scenario(7, i, n_agents=64, n_polylines=512) for i in range(n_scn)]
model = None
launches = [0]
engines = {}
if pollute:
    sys.path.insert(0, 'tests')
    from gpu_utils import Polluter

def run(lanes, p2, tail, cached):
    global model
    eng = engines.get(lanes)                    # one engine per lane count, re-used (reset) by every run
    if eng is None:
        eng = RolloutEngine(cfg, w, 'cuda:0', max_ctx=max(64, 16 * n_scn), seed=3, model=model, lanes=lanes, compact=not f32, contacts=contacts)
        for key, ch in zip((0, 1, 3), sel):
            eng.lib.ctrlsim_set_option(key, int(ch))
        model = eng.model
        eng.pass2_on_side, eng.tail_on_side, eng.cached_on_side = p2, tail, cached
        eng.forward_waits_for_sim = guard
        eng.load_scenarios(scns, steps=90)
        if delay_us and lanes > 1:
            sys.path.insert(0, 'tests')
            import gpu_utils
            gpu_utils.delay_simulator_steps(eng, delay_us)
        engines[lanes] = eng
    else:
        eng.reset()
    if pollute and lanes > 1:
        with Polluter() as pol:
            r = eng.run(90).results()
        launches[0] += pol.launches
    else:
        r = eng.run(90).results()
    return r["tokens"].copy(), r["states"].copy(), r["coll"].copy()


ref = run(1, False, False, False)
bad = 0
for k in range(runs):
    b = run(2, p2, tail, cached)
    nt = int((ref[0] != b[0]).sum())
    ds = float(np.abs(ref[1] - b[1]).max())
    bad += nt > 0 or ds > 0
    print(f"run {k}: token differences {nt}, max |state difference| {ds}", flush=True)
    if nt > 0 or ds > 0:
        # where: (scenario, vehicle, step, field) of the EARLIEST differing state rows, fields x y vx vy heading length width exist
        w = np.argwhere(ref[1] != b[1])
        w = w[np.argsort(w[:, 2], kind="stable")]
        t0_ = int(w[0, 2])
        print("   first differing step", t0_, ": (scn, veh) -> fields:",
              {(int(s_), int(v_)): sorted(int(f) for f in w[(w[:, 2] == t0_) & (w[:, 0] == s_) & (w[:, 1] == v_), 3])
               for s_, v_ in {(int(q[0]), int(q[1])) for q in w[w[:, 2] == t0_]}}, flush=True)
        w = w[:12]
        print("   first state differences (scn, veh, t, field: ref -> got):",
              [(int(s_), int(v_), int(t_), int(f_), float(ref[1][s_, v_, t_, f_]), float(b[1][s_, v_, t_, f_])) for s_, v_, t_, f_ in w], flush=True)
        tw = np.argwhere(ref[0] != b[0])
        cw = np.argwhere(ref[2] != b[2])
        print("   earliest token difference (scn, veh, t):", tw[np.argsort(tw[:, 2], kind='stable')][:3].tolist() if len(tw) else None,
              " earliest collision-flag difference:", cw[np.argsort(cw[:, 2], kind='stable')][:3].tolist() if len(cw) else None,
              " steps with any state difference:", sorted(set(np.argwhere(ref[1] != b[1])[:, 2].tolist()))[:20], flush=True)
if pollute:
    print(f"{launches[0]} polluter launches")
print(f"{bad} of {runs} runs differ from the single-stream rollout (p2={int(p2)} tail={int(tail)} cached={int(cached)})")
