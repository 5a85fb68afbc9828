"""BASELINE.md 3.2 / SURVEY 8(d): is the CPU port (oracle/rollout_oracle.py + oracle/sim_oracle.c — what bench.py times as `cpu_baseline`,
kind "port", because /root/reference cannot travel to the GPU box) a fair stand-in for the true reference?  Run in the BUILD CONTAINER
(needs /root/reference): the same headline-shaped scene (64 vehicles x 512 polylines, full model) is rolled for a few steps by the
UNMODIFIED reference policy + real FreeCar / Box2D (oracle/gen_golden.py::ref_closed_loop) and by the port, on the same host threads;
prints seconds per focal-group-step of both.  Usage: python tools/cpu_port_vs_reference.py [steps=3] [threads=all]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch

import ctrlsim_amd  # noqa: F401
from ctrlsim_amd import spec, weights, scenarios
import gen_golden
import rollout_oracle
import sim_libs

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
torch.set_num_threads(threads)
cfg = spec.make_cfg(nocturne__steps=max(steps, 40))      # (the reference sizes its buffers by nocturne.steps: a full window)
d = spec.Dims(cfg)
w = weights.generate(d, 0)
scn = scenarios.make_scenario(0, 5, n_agents=64, n_polylines=512, n_points=d.NP, extent=100.0)
sim_libs.build_oracle()

t0 = time.perf_counter()
r = gen_golden.ref_closed_loop(cfg, w, scn, steps, seed=9)
t_ref = time.perf_counter() - t0
g_ref = int(r["n_groups"].sum())

import model_oracle
model_oracle.FUSED_SDPA = True          # the reference's attention op (bench.py's cpu_baseline sets it too)
ro = rollout_oracle.RolloutOracle(cfg, w, seed=9, threads=threads)
t0 = time.perf_counter()
o = ro.run(scn, steps, sim_libs.OracleSim, dense_window=True)      # full T-step windows, as the reference forwards them
t_port = time.perf_counter() - t0
g_port = int(o["n_groups"].sum())
same = np.array_equal(o["tokens"], r["tokens"])
cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
print(f"host: {cpu}, {threads} threads, torch {torch.__version__}")
print(f"reference (unmodified policy + real FreeCar/Box2D): {t_ref:.1f} s for {steps} steps, {g_ref} focal-group steps -> {t_ref / g_ref:.3f} s per focal-group step, "
      f"{64 * steps / t_ref:.2f} agent-steps/s")
print(f"port (oracle/rollout_oracle.py + sim_oracle.c):      {t_port:.1f} s for {steps} steps, {g_port} focal-group steps -> {t_port / g_port:.3f} s per focal-group step, "
      f"{64 * steps / t_port:.2f} agent-steps/s")
print(f"tokens identical: {same}; port / reference time per focal-group step: {t_port / g_port / (t_ref / g_ref):.2f}")
