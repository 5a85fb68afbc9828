#!/usr/bin/env python
"""bench.py — closed-loop rollout throughput of the HIP path (agent-steps/s), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2]: S = 2048 synthetic Waymo-shaped scenarios PER GPU (64 vehicles, 512 road polylines x
100 points, 90 simulator steps, CtRL-Sim base model with random-init weights), all resident in HBM before the timed
region (uploaded once, untimed).  One bench "step" = one complete 90-step closed-loop rollout of ONE SLICE of that
resident batch: policy inference (focal grouping, context build, two-pass transformer, sampling) + simulator step +
collision flags + history update.  The K timed steps roll out K consecutive slices that cover the 2048 scenarios
EXACTLY ONCE (slice sizes floor/ceil(S/K)), so value = S x 64 x 90 x n_gpus / (time of the K steps) is the
configs[2] figure itself; the W warm-up steps roll out slices of the same size from the same batch.  Scenarios are
independent, so ranks shard them with no data-path collective (weak scaling: S per GPU fixed); the only collective is
one all-reduce of the metric accumulators after the rollouts.

Printed by rank 0: ONE JSON line on stdout — the LAST line of stdout, < 4 KB (SHORT_LINE_LIMIT): metric, value = whole-job
agent-steps/s, roofline of the dominant KERNEL (HIP events on the launch stream during the timed region), cpu_baseline = the
CPU oracle timed on this box's host cores on a bounded sample, the parity spot check, the sampled shader clock / socket power.
Everything else — per-kernel rows of both streams, the class view, satellites against the HBM roof, per-class attention rows,
phases, per-rank times, notes — goes to the detail file (--detail-file, default bench_detail.json) and, prefixed, to stderr.
`python bench.py --gpus N` without a launcher starts the N ranks itself.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_16BIT_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA (AMD's 5 PF figure includes 2:1 sparsity)
PEAK_HBM_TBPS = 8.0               # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s measured streaming copy)
# Both matrix kernels evaluate every fp32 product as NPROD 16-bit partial products with fp32-class accuracy (csrc/split.h: three
# fp16 products of two-plane splits, or six bf16 products of three-plane splits): the roof of the ALGORITHMIC (fp32-equivalent)
# FLOP rate is the dense 16-bit MFMA peak / NPROD.
CLASS_KEYS = ("gemm", "attention", "build_context", "assemble_tokens", "sim_step", "map_pool")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8, help="timed steps = slices that cover the resident batch exactly once")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scenarios", type=int, default=2048, help="scenarios resident per GPU (BASELINE configs[2]: 2048)")
    ap.add_argument("--agents", type=int, default=64)
    ap.add_argument("--polylines", type=int, default=512)
    ap.add_argument("--rollout-steps", type=int, default=90)
    ap.add_argument("--context-slots", type=int, default=0,
                    help="NON-REFERENCE secondary of SURVEY.md 8(d): agent slots per context (cfg.dataset.waymo.max_num_agents; 0 = the reference's 24). "
                         "64 = one wide context per 64-vehicle scenario instead of three focal groups")
    ap.add_argument("--context-polylines", type=int, default=0, help="with --context-slots: polylines per context (0 = the reference's 200)")
    ap.add_argument("--max-ctx", type=int, default=1024, help="model batch (contexts per forward chunk; the two lanes' workspaces take 2 x 84 GB of the 288 GB at 1024)")
    ap.add_argument("--lanes", type=int, default=2, help="scenario sets in flight per GPU (engine.py)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tilt", type=float, nargs=3, default=(0.0, 0.0, 0.0))
    ap.add_argument("--tilt-sweep", action="store_true",
                    help="BASELINE configs[4]: scenario i runs with goal = veh = road tilt TILT_SWEEP[i %% 8] (one batch)")
    ap.add_argument("--sizes", type=str, default="", help="A/B only: explicit context size classes, e.g. 6,8,10,12,14,16,20,24 (round 2's set)")
    ap.add_argument("--side", type=str, default=None,
                    help="few-row kernels on the lanes' side streams: comma list of p2, tail, cached ('' = none; default: the engine's)")
    ap.add_argument("--sim-guard", action="store_true", help="A/B: the round-2 stream guard (a forward pass waits for pending simulator steps)")
    ap.add_argument("--pipeline", action="store_true",
                    help="A/B: pipelined lanes (engine.run_jobs: a lane's K/V-cached steps underneath the other lanes' full-recompute steps; with "
                         "--lanes 3 the third lane rolls the next job's cached steps ahead) instead of one run() per slice with both lanes "
                         "starting at t = 0 together.  Measured in round 4: no gain (profiles/README.md) — the default stays one run() per slice")
    ap.add_argument("--no-class-profile", action="store_true", help="skip the untimed extra slice with per-class attention cycle accounting")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-samples", action="store_true",
                    help="do not sample shader clock / socket power (rocm-smi, a host thread, every few seconds) during the timed region")
    ap.add_argument("--spot-check", type=int, default=4, help="scenarios re-rolled alone after the timed region (bit-identity check; 0 = off)")
    ap.add_argument("--cpu-sample-scenarios", type=int, default=4)
    ap.add_argument("--cpu-sample-steps", type=int, default=4)
    ap.add_argument("--cpu-sample-budget-s", type=float, default=60.0,
                    help="time box of the CPU sample: no further scenario is started once this much CPU time is spent (at least one runs)")
    ap.add_argument("--detail-file", type=str, default=os.path.join(ROOT, "bench_detail.json"),
                    help="everything that does not fit the one short stdout line: per-kernel rows, satellites, per-class attention, notes")
    ap.add_argument("--fallback-slice", type=int, default=1,
                    help="untimed: roll the first slice once more on the bf16x6 operand split (the automatic fallback of a checkpoint that "
                         "trips the fp16 range guard) and report its throughput in the detail file (0 = off)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the MULTI-RANK FLOW only (tests/test_dist_cpu.py: world 8 over gloo, no GPU): a stand-in engine that "
                         "sleeps instead of rolling, everything rank-dependent — scenario ids, tilt per global id, a model batch reduced on some "
                         "ranks, barriers, the MAX / gather of the elapsed time, the SUM all-reduce of the metric vector, the parting barrier, "
                         "rank 0's CPU sample after the others left — through the same code as a real run.  The line it prints is marked "
                         "dry_run and is never a measurement")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: --gpus MEANS N.  Re-exec this same command line as N ranks, one per GPU, through
        # torch.distributed.run on 127.0.0.1 (rank r -> LOCAL_RANK r -> cuda:r); rank 0's stdout line is this process's stdout line.
        return _self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: launch with --nproc-per-node {args.gpus} "
                         f"(or plain `python bench.py --gpus {args.gpus}`, which launches the ranks itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    dry = bool(args.dry_run)
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # debugging aid only: several ranks on ONE GPU with gloo collectives, to exercise the multi-rank flow on a 1-GPU box
    shared_gpu = os.environ.get("CTRLSIM_BENCH_DEBUG_SHARED_GPU") == "1" and not dry
    if shared_gpu:
        local_rank %= torch.cuda.device_count()
    # CTRLSIM_BENCH_FORCE_DIST=1: initialise torch.distributed (backend nccl = RCCL) even at world size 1, so that the job's one
    # collective and its barriers run through RCCL on a single-GPU box exactly as they do on 8 (tests/test_gpu_dist.py)
    force_dist = os.environ.get("CTRLSIM_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(port)), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        if shared_gpu or dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    device = "cpu" if dry else f"cuda:{local_rank}"
    coll_device = "cpu" if (shared_gpu or dry) else device
    if not dry:
        torch.cuda.set_device(device)

    import ctrlsim_amd  # noqa: F401
    from ctrlsim_amd import spec, weights, scenarios, metrics
    from ctrlsim_amd.dist import rank_plan, gather_scalar
    if not dry:
        from ctrlsim_amd import _lib
        from ctrlsim_amd.engine import RolloutEngine
        for kv in filter(None, os.environ.get("CTRLSIM_OPTIONS", "").split(",")):   # "<option>=<value>,...": kernel A/B runs only
            k, v = kv.split("=")
            _lib.lib().ctrlsim_set_option(int(k), int(v))
    over = {}
    if args.context_slots:
        over["dataset__waymo__max_num_agents"] = args.context_slots
    if args.context_polylines:
        over["dataset__waymo__max_num_road_polylines"] = args.context_polylines
    cfg = spec.make_cfg(nocturne__steps=args.rollout_steps, nocturne__history_steps=1, **over)
    d = spec.Dims(cfg)
    w = weights.generate(d, 0)
    S, N, R, K = args.scenarios, args.agents, args.rollout_steps, args.steps
    if K < 1 or K > S:
        raise SystemExit("--steps must be in [1, --scenarios]: the timed steps are slices of the resident batch")
    # global scenario ids: interleaved over ranks (rank r takes r, r+W, ...) so results do not depend on W; with --tilt-sweep the tilt
    # of a scenario follows its GLOBAL id (ctrlsim_amd/dist.py: rank_plan — the same function the CPU rehearsal of world 8 runs)
    ids, sweep_tilt = rank_plan(rank, world, S, args.tilt_sweep)
    t_gen = time.perf_counter()
    scns = scenarios.make_batch(args.seed, ids, n_agents=N, n_polylines=args.polylines)
    gen_s = time.perf_counter() - t_gen
    tilt = tuple(args.tilt)
    if args.tilt_sweep:                                       # the sweep values of config 5
        tilt = sweep_tilt
    if os.environ.get("CTRLSIM_MAIN_CU_MASK") and not dry:                # A/B experiment: the main stream on a subset of the compute units
        from ctrlsim_amd.engine import _new_stream
        torch.cuda.set_stream(_new_stream(torch.device(device), os.environ["CTRLSIM_MAIN_CU_MASK"]))
    eng = None
    max_ctx_asked = args.max_ctx
    while eng is None:                                        # the lanes' workspaces are sized for max_ctx plain contexts: if the
        try:                                                  # device does not have that much free, halve the model batch
            if dry:
                eng = _DryEngine(cfg, rank, ids, args.max_ctx)
            else:
                eng = RolloutEngine(cfg, w, device, max_ctx=args.max_ctx, seed=args.seed, tilt=tilt, lanes=args.lanes,
                                    sizes=tuple(int(x) for x in args.sizes.split(",")) if args.sizes else None)
        except torch.OutOfMemoryError:
            if args.max_ctx <= 128:
                raise
            args.max_ctx //= 2
            print(f"[bench] out of device memory: model batch reduced to {args.max_ctx} contexts", file=sys.stderr)
        if eng is None and not dry:                           # (outside the handler: the traceback no longer pins the half-built engine)
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    if args.sim_guard:
        eng.forward_waits_for_sim = True
    if args.side is not None:
        on = set(filter(None, args.side.split(",")))
        eng.pass2_on_side, eng.tail_on_side, eng.cached_on_side = "p2" in on, "tail" in on, "cached" in on
    sync = (lambda: None) if dry else torch.cuda.synchronize
    sync()
    t_up = time.perf_counter()
    eng.load_scenarios(scns, steps=R)                         # host -> HBM: the only PCIe traffic of a rollout (untimed)
    sync()
    upload_ms = (time.perf_counter() - t_up) * 1e3
    lib = None if dry else _lib.lib()
    cuts = [S * i // K for i in range(K + 1)]                 # K slices, sizes floor / ceil (S / K), sum = S

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    def bench_step(i):                                        # one 90-step closed-loop rollout of slice i
        a, b = cuts[i % K], cuts[i % K + 1]
        eng.reset(a, b)
        eng.run(R, s0=a, s1=b)                                # (finite-logit guard: after the timed region, below)

    def bench_steps(first, n):
        """Bench steps first .. first + n - 1.  Pipelined (default): the lanes take half-slices as they come free and stay about half a
        rollout apart (engine.run_jobs), so the steps overlap in time; every slice is still reset and rolled exactly once."""
        if not args.pipeline:
            for i in range(first, first + n):
                bench_step(i)
            return
        jobs = []
        for i in range(first, first + n):
            a, b = cuts[i % K], cuts[i % K + 1]
            m = max(1, min(args.lanes, b - a))
            jobs += [(a + (b - a) * j // m, a + (b - a) * (j + 1) // m) for j in range(m)]
        eng.run_jobs(jobs, R)

    bench_steps(0, args.warmup)
    barrier()
    if not dry:
        lib.ctrlsim_prof_enable(1)
    eng.record_phases, eng.phase_events = not args.pipeline, []
    eng.full_pass_contexts = np.zeros(len(eng.sizes), np.int64)
    clocks = _ClockSampler(local_rank) if (rank == 0 and not dry and not args.no_clock_samples) else None
    t0 = time.perf_counter()
    bench_steps(0, K)
    t_own = time.perf_counter() - t0                          # this rank's own time to its last kernel (before it waits for the others)
    barrier()
    elapsed = time.perf_counter() - t0
    clock_samples = clocks.stop() if clocks else None
    eng.record_phases = False
    cached_s, sliding_s = eng.phase_times()
    if dry:
        return _finish_dry(args, dist, rank, world, eng, cfg, scns, ids, elapsed, t_own, max_ctx_asked, coll_device, gather_scalar, metrics, tilt)
    ncls = int(lib.ctrlsim_prof_classes())
    # Launch intervals (HIP events on the launch stream).  The engine runs the big kernels of both lanes back to back on ONE
    # stream (the current one) and, underneath them on the lanes' side streams, the few-row kernels of the second pass, the
    # simulator step and the grouping: intervals on different streams overlap in time, so the per-class rates below are taken
    # from the launches of the main stream (> 98 % of the FLOPs) and the side-stream launches are reported next to them.
    nsub = int(lib.ctrlsim_prof_subclasses())
    kms = (C.c_double * nsub)(); kcnt = (C.c_int64 * nsub)(); kfl = (C.c_double * nsub)(); kby = (C.c_double * nsub)()
    ms = (C.c_double * ncls)(); cnt = (C.c_int64 * ncls)(); fl = (C.c_double * ncls)(); by = (C.c_double * ncls)()
    sms = (C.c_double * ncls)(); scnt = (C.c_int64 * ncls)(); sfl = (C.c_double * ncls)(); sby = (C.c_double * ncls)()
    main_stream = _lib.stream_ptr()
    _lib.check(lib.ctrlsim_prof_collect_stream(main_stream, 1, ms, cnt, fl, by), "prof_collect")
    _lib.check(lib.ctrlsim_prof_collect_stream(main_stream, 0, sms, scnt, sfl, sby), "prof_collect")
    _lib.check(lib.ctrlsim_prof_collect_sub(main_stream, 1, kms, kcnt, kfl, kby), "prof_collect_sub")
    skms = (C.c_double * nsub)(); skcnt = (C.c_int64 * nsub)(); skfl = (C.c_double * nsub)(); skby = (C.c_double * nsub)()
    _lib.check(lib.ctrlsim_prof_collect_sub(main_stream, 0, skms, skcnt, skfl, skby), "prof_collect_sub")
    for i in range(2, ncls):                                  # satellites: wherever they ran
        ms[i] += sms[i]; cnt[i] += scnt[i]; fl[i] += sfl[i]; by[i] += sby[i]
    lib.ctrlsim_prof_enable(0)
    elapsed, rank_times, rank_ctx = _reduce_times(dist, elapsed, t_own, args.max_ctx, coll_device, gather_scalar)

    # ---- parity spot check (untimed): a few scenarios of the batch are rolled again, ALONE, by a second engine (one lane, small model
    # batches, no other scenario in their forward chunks): tokens, RTG bins, collision flags and float32 trajectories must be
    # BIT-identical to what the timed rollout left in HBM — a regression of the batching / lane / class machinery shows here
    spot = None
    if rank == 0 and args.spot_check > 0:
        pick = sorted({int(x) for x in np.linspace(0, S - 1, args.spot_check)})
        tl = tilt[pick] if isinstance(tilt, np.ndarray) else tilt
        e2 = RolloutEngine(cfg, w, device, max_ctx=64, seed=args.seed, tilt=tl, lanes=1, model=eng.model)
        e2.load_scenarios([scns[i] for i in pick], steps=R)
        r2 = e2.rollout(R)
        torch.cuda.synchronize()
        ix = torch.tensor(pick, device=device)
        same = {k: bool(torch.equal(getattr(eng, k)[ix], getattr(e2, k))) for k in ("hist_tok", "hist_rtg", "coll")}
        dstate = float((eng.hist_states[ix] - e2.hist_states).abs().max().item())
        spot = {"scenarios_rerolled_alone": [ids[i] for i in pick], "tokens_identical": same["hist_tok"], "rtg_bins_identical": same["hist_rtg"],
                "collision_flags_identical": same["coll"], "max_abs_state_difference": dstate,
                "identical": all(same.values()) and dstate == 0.0,
                "note": "second engine, one lane, the scenarios alone in their model batches vs the timed two-lane multi-class rollout"}
        del e2, r2
        eng._bind()
        if not spot["identical"]:
            # a run-time defence, not a report: a rollout that depends on what shares its model batch / lanes / streams is wrong
            # (the co-residency hazard of DESIGN.md section 4 showed exactly like this) and must not produce a bench line
            print(json.dumps({"parity_spot_check": spot}), file=sys.stderr)
            raise RuntimeError("parity spot check failed: scenarios re-rolled alone differ from the timed rollout")

    # ---- causal self-attention by context size class (untimed): the first slice is rolled once more with the kernel's per-class cycle
    # accounting on (one atomic per workgroup: not inside the timed region) — where the largest single kernel spends its time
    attn_by_class = None
    full_ctx = np.asarray(eng.full_pass_contexts, np.int64).copy()
    if rank == 0 and not args.no_class_profile:
        a, b = cuts[0], cuts[1]
        _lib.check(lib.ctrlsim_attn_class_prof(1, None), "attn_class_prof")
        eng.reset(a, b)
        eng.run(R, s0=a, s1=b)
        torch.cuda.synchronize()
        buf = (C.c_uint64 * 64)()
        _lib.check(lib.ctrlsim_attn_class_prof(0, buf), "attn_class_prof")
        eng._unchecked = eng._unchecked[:-1]                          # (the extra roll is not part of what check_finite may repeat)
        tot = float(sum(buf[2 * s] for s in range(32))) or 1.0
        T_ = d.T
        attn_by_class = []
        for k, A_ in enumerate(eng.sizes):
            cyc, wgs = int(buf[2 * A_]), int(buf[2 * A_ + 1])
            if wgs == 0:
                continue
            rep = 1 if A_ < d.A else 0
            Ar = A_ - rep
            A3 = 3.0 * Ar
            pairs = A3 * A3 * T_ * (T_ - 1) / 2.0 + T_ * Ar * (3.0 * Ar + 3.0)
            if rep:
                pairs += A3 * (3.0 * T_ * (T_ - 1) / 2.0 + T_) + 3.0 * A3 * T_ * (T_ - 1) / 2.0 + 3.0 * Ar * T_ + 9.0 * T_ * (T_ - 1) / 2.0 + 6.0 * T_
            attn_by_class.append({"context_slots": int(A_), "rows_per_context": int(3 * T_ * A_), "workgroups": wgs,
                                  "share_of_workgroup_cycles": cyc / tot, "visible_pairs_per_context_and_head": pairs,
                                  "contexts_in_timed_full_passes": int(full_ctx[k]),
                                  "cycles_per_1e6_visible_pairs": cyc / max(wgs / ((3 * T_ * A_ + 127) // 128), 1) / pairs * 1e6})
        eng._bind()

    # ---- metrics: the evaluator's accumulators are built on the device (ctrlsim_metrics_pack: rank-side work independent of S)
    # and combined by ONE all-reduce of that ~10 KB vector — the only collective of the job.  The synthetic
    # scenes' "log" is the constant-velocity continuation of the initial state.
    res = {"n_groups": eng.groups_per_step}
    f64 = lambda a: np.stack([np.asarray(getattr(s, a), np.float64) for s in scns])
    x0, y0, hd, sp = f64("x"), f64("y"), f64("heading"), f64("speed")
    tt = np.arange(R + 1)[None, None, :] * cfg.nocturne.dt
    gt = np.stack([x0[..., None] + (sp * np.cos(hd))[..., None] * tt, y0[..., None] + (sp * np.sin(hd))[..., None] * tt,
                   np.broadcast_to(hd[..., None], (S, N, R + 1)), np.broadcast_to(sp[..., None], (S, N, R + 1)),
                   np.ones((S, N, R + 1))], -1)
    goals4 = np.concatenate([f64("goal_pos"), f64("goal_heading")[..., None], f64("goal_speed")[..., None]], -1)
    vec = eng.metrics_pack(gt, goals4)
    torch.cuda.synchronize()
    if eng.nonfinite(reset=False):
        raise FloatingPointError("guard events during the rollout (csrc/split.h: activation range; csrc/sim.hip: contact table)")
    vec = vec.to(coll_device)
    if dist is not None:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    acc = metrics.MetricAccumulators().unpack(vec.cpu().numpy())
    n_vec = int(vec.numel())
    backend = dist.get_backend() if dist is not None else "none (single process)"
    if dist is not None:                      # every collective of the job is done: the ranks part here, rank 0 goes on to the (untimed)
        dist.barrier()                        # CPU baseline and the report without holding seven idle ranks in a process group
        dist.destroy_process_group()
        dist = None

    if rank == 0:
        agent_steps = S * N * R * world
        value = agent_steps / elapsed
        ctx_per_rollout = int(res["n_groups"].sum())
        dom = 0 if ms[0] >= ms[1] else 1
        f16 = eng.scheme == 1
        nprod, scheme = (3, "two fp16 planes, 3") if f16 else (6, "three bf16 planes, 6")
        PEAK_FP32_EQUIV_TFLOPS = PEAK_16BIT_MFMA_TFLOPS / nprod
        names = ("inproj_rs_kernel + gemm_ws256_kernel + gemm_nt_bf16x6_kernel + ffn_fused_bf16x6_kernel (every nn.Linear incl. fused LayerNorm / K-V image epilogues and the fused "
                 f"feed-forward block; split-operand MFMA 32x32x16: {scheme} partial products per fp32 product)",
                 "attention_bf16x6_kernel (all multi-head attention; split-operand MFMA flash attention, structured mask)")

        # HBM bytes per launch from the PMC counters: rocprofv3 cannot run inside this process, so they come from separate --pmc passes over
        # a smaller run of this same command (tools/pmc_traffic.sh), committed under profiles/ PER KERNEL with ONE counter convention for
        # every file of the round: FETCH_SIZE x 2 (gfx950 tallies its 128-byte read requests at 64 B — calibrated in round 5 on known-byte
        # launches of every access pattern and of each hot kernel: profiles/r05_pmc_calibration.json) + WRITE_SIZE x 1.
        pmc, pmc_src = {}, None
        for cand in ("r06_pmc_traffic.json", "r05w_pmc_traffic.json", "r05_pmc_traffic.json"):
            pmc_path = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(pmc_path):
                pmc, pmc_src = json.load(open(pmc_path)), "profiles/" + cand
                break
        KIND_KEYS = ("other", "linear_kv_images", "linear_residual_layernorm", "linear_plain", "ffn_fused", "attention_causal", "attention_keypad")
        CLASS_KINDS = ((0, 1, 2, 3, 4), (5, 6))             # kinds of the Linear class / the attention class

        def kind_ratio(kind):
            """measured HBM bytes / algorithmic bytes of kernel kind `kind` in the PMC run (same kernels and shapes per context, smaller
            launches: counter passes serialise the kernels); None when the profile has no entry for it."""
            e = pmc.get("kernels", {}).get(KIND_KEYS[kind])
            return e.get("hbm_over_algorithmic") if e else None

        def kind_bytes(kind, main_only=True):
            """algorithmic bytes and launches of a kind in THIS run (main stream; full-row + few-row launches)."""
            b = kby[2 * kind] + kby[2 * kind + 1] + (0.0 if main_only else skby[2 * kind] + skby[2 * kind + 1])
            n = kcnt[2 * kind] + kcnt[2 * kind + 1] + (0 if main_only else skcnt[2 * kind] + skcnt[2 * kind + 1])
            return b, n

        def traffic(i, alg_per_launch):
            """HBM bytes per launch of class i: every kind's measured-over-algorithmic ratio applied to THIS run's algorithmic bytes of
            that kind, summed over the class's kinds, per launch of the class."""
            tot_alg = tot_meas = 0.0
            for kind in CLASS_KINDS[i]:
                b, _ = kind_bytes(kind)
                r = kind_ratio(kind)
                if b <= 0:
                    continue
                if r is None:
                    return None
                tot_alg += b
                tot_meas += r * b
            return (tot_meas / tot_alg) * alg_per_launch if tot_alg > 0 else None

        def traffic_measured(i):
            return {KIND_KEYS[k]: pmc.get("kernels", {}).get(KIND_KEYS[k]) for k in CLASS_KINDS[i] if KIND_KEYS[k] in pmc.get("kernels", {})} or None

        def cls(i):
            a = fl[i] / (ms[i] * 1e-3) / 1e12 if ms[i] > 0 else None
            n = max(cnt[i], 1)
            return {"kernel": names[i], "achieved": a, "peak": PEAK_FP32_EQUIV_TFLOPS, "unit": "TFLOP/s",
                    "frac": a / PEAK_FP32_EQUIV_TFLOPS if a else None, "avg_launch_ms": ms[i] / n,
                    "launches": int(cnt[i]), "time_share_of_step": ms[i] * 1e-3 / elapsed,
                    "mfma_executed_tflops": nprod * a if a else None, "mfma_peak_tflops": PEAK_16BIT_MFMA_TFLOPS,
                    "algorithmic_flops_per_launch": fl[i] / n, "algorithmic_hbm_bytes_per_launch": by[i] / n,
                    "traffic": traffic(i, by[i] / n), "traffic_source": pmc_src, "traffic_convention": pmc.get("_convention"),
                    "traffic_measured": traffic_measured(i),
                    "hbm_rate_at_algorithmic_bytes_TBps": by[i] / (ms[i] * 1e-3) / 1e12 if ms[i] > 0 else None,
                    "side_stream": {"launches": int(scnt[i]), "event_ms_total": sms[i], "flop_share": sfl[i] / max(fl[i] + sfl[i], 1.0),
                                    "algorithmic_hbm_bytes_total": sby[i],
                                    "note": "few-row launches of the second pass, run on the lanes' side streams underneath the "
                                            "main stream's kernels (their intervals overlap those above and are not in them)"}}

        def sat(i):                                           # satellite kernels: algorithmic bytes / event time vs the HBM roof
            if cnt[i] == 0 or ms[i] <= 0:
                return None
            rate = by[i] / (ms[i] * 1e-3) / 1e12
            row = {"achieved": rate, "peak": PEAK_HBM_TBPS, "unit": "TB/s", "frac": rate / PEAK_HBM_TBPS,
                   "avg_launch_ms": ms[i] / cnt[i], "launches": int(cnt[i]), "time_share_of_step": ms[i] * 1e-3 / elapsed,
                   "algorithmic_hbm_bytes_per_launch": by[i] / cnt[i]}
            mp_path = os.path.join(ROOT, "profiles", "r06_c_pmc_map_pool.json")
            if CLASS_KEYS[i] == "map_pool" and os.path.exists(mp_path):
                # map_pool is a VECTOR-pipe kernel (no HBM roof: it moves its bytes once, at 0.1 TB/s): its roof is the vector pipe's time.
                # Busy SIMD-cycles per polyline from the SQ counter passes (SQ_ACTIVE_INST_VALU, profiles/r06_c_pmc_map_pool.md: a wave64
                # v_fma_f32 holds a SIMD's vector pipe for 2 cycles, a v_pk_fma_f32 for 4), time from this run's HIP events.
                mp = json.load(open(mp_path))
                polylines = fl[i] / 1.5e6                          # launch_map_pool tags 1.5 MFLOP per polyline
                busy_rate = polylines * mp["valu_busy_simd_cycles_per_polyline"] / (ms[i] * 1e-3)
                peak_rate = mp["simds"] * 2.4e9                    # 1024 SIMDs x the 2.4 GHz maximum clock
                row["vector_pipe"] = {"bound": "valu", "achieved": busy_rate / 1e12, "peak": peak_rate / 1e12, "unit": "T busy SIMD-cycles/s",
                                      "frac": busy_rate / peak_rate, "kernel": mp["kernel"], "valu_insts_per_polyline": mp["valu_insts_per_polyline"],
                                      "active_lanes_per_instruction": mp["active_lanes_per_valu_inst"],
                                      "counter_frac_at_profiled_clock": mp["valu_busy_frac_measured"], "source": "profiles/r06_c_pmc_map_pool.md",
                                      "note": "peak at the 2.4 GHz maximum clock; the counters' own cycle count (profiled run) gives "
                                              "counter_frac_at_profiled_clock = 0.87: the kernel is at its FMA count (packed FMAs, round 6)"}
            return row
        KNAMES = ("other", "Linear + K/V-image epilogue (QKV, memory K/V): inproj_rs_kernel (row-stationary; weight-stationary gemm_ws256_kernel<..,KV> / tiled gemm_nt_bf16x6_kernel<2,2,2,..,KVIMG> by option or when K != 256)",
                  "Linear + residual + LayerNorm epilogue (attention out-projections, MLP layers): gemm_ws256_kernel<..,LN> (weight-stationary; tiled gemm_nt_bf16x6_kernel<1,4,2,..,LN> when K != 256)",
                  "plain Linear (cross-attention query projection, heads, map / embedding layers): gemm_ws256_kernel (256 -> 256) / gemm_nt_bf16x6_kernel<2,2,2>",
                  "fused feed-forward block: ffn_fused_bf16x6_kernel",
                  "causal self-attention: attention_bf16x6_kernel<1,true,true> (masks from the per-class table; <1,true,false> for the few-row launches)",
                  "key-padded scene / cross attention: attention_bf16x6_kernel<0,true>")

        def kernel_rows(kms=kms, kcnt=kcnt, kfl=kfl, kby=kby):
            """Launches by kernel (2 * kind + few-row flag, include/ctrlsim.h: ctrlsim_prof_collect_sub): main stream by default."""
            rows = []
            for i in range(nsub):
                if kcnt[i] == 0 or kms[i] <= 0:
                    continue
                a = kfl[i] / (kms[i] * 1e-3) / 1e12
                r = kind_ratio(i // 2)
                rows.append({"kernel": KNAMES[i // 2] + (" — few-row launches (last layer on the queried rows, second pass, cached steps)" if i % 2 else ""),
                             "kind": KIND_KEYS[i // 2],
                             "achieved": a, "frac": a / PEAK_FP32_EQUIV_TFLOPS, "unit": "TFLOP/s", "avg_launch_ms": kms[i] / kcnt[i],
                             "launches": int(kcnt[i]), "time_share_of_step": kms[i] * 1e-3 / elapsed,
                             "algorithmic_hbm_bytes_per_launch": kby[i] / kcnt[i],
                             "traffic": None if r is None else r * kby[i] / kcnt[i],
                             "hbm_rate_at_algorithmic_bytes_TBps": kby[i] / (kms[i] * 1e-3) / 1e12})
            return sorted(rows, key=lambda r: -r["time_share_of_step"])
        e2e = (sum(fl[i] for i in range(ncls)) + sfl[0] + sfl[1]) / elapsed / 1e12
        roof = {"bound": "mfma", **cls(dom),
                "note": "achieved = algorithmic fp32 FLOPs (2MNK per Linear; 128 per visible (q,k) pair and head) / summed "
                        f"HIP-event time of the class; peak = dense 16-bit MFMA peak / {nprod} because each fp32 product costs {nprod} "
                        f"MFMA products ({scheme} products, fp32-class accuracy: csrc/split.h); the f32-input MFMA path (157.3 TF peak) is "
                        "selectable per engine (ctrlsim_bind_options); traffic = HBM bytes per launch: per KERNEL (2 x FETCH_SIZE + WRITE_SIZE) / algorithmic "
                        "bytes measured by separate rocprofv3 --pmc passes over a smaller run of this workload (traffic_measured, committed "
                        "under profiles/ with the calibration of the counter factors) x this run's algorithmic bytes of that kernel",
                "mfma_sustained_measured": _sustained(nprod, ms, fl, dom),
                "other": cls(1 - dom),
                "satellite": {CLASS_KEYS[i]: sat(i) for i in range(2, ncls)},
                "satellite_note": "HBM-side kernels (SURVEY 8d): algorithmic bytes (DESIGN.md 4) / HIP-event time vs the 8 TB/s HBM "
                                  "peak; sim_step is latency-bound (one workgroup per scenario, alone on its CU) and runs on the lane's "
                                  "side stream " + ("underneath the other lane's grouping / context kernels — with --sim-guard a forward pass "
                                                    "waits for every pending simulator step, so it never runs beside matrix kernels"
                                                    if args.sim_guard else
                                                    "underneath the other lane's kernels, matrix kernels included (the co-residency hazard that "
                                                    "forbade this in round 2 is gone from the build: DESIGN.md section 4)"),
                "kernels": kernel_rows(),
                "kernel_bytes_all_streams": {KIND_KEYS[k]: {"launches": int(kind_bytes(k, False)[1]), "algorithmic_hbm_bytes": kind_bytes(k, False)[0]}
                                             for k in range(1, len(KIND_KEYS)) if kind_bytes(k, False)[1]},
                "causal_attention_by_size_class": attn_by_class,
                "causal_attention_by_size_class_note": "untimed extra roll of the first slice with ctrlsim_attn_class_prof: share of the causal "
                                                        "kernel's workgroup cycles (full-row launches) per context size class; cycles per 1e6 visible "
                                                        "(query, key) pairs and head = the per-class efficiency (lower is better; the plain 24-slot "
                                                        "class is the reference)",
                "kernels_on_side_streams": kernel_rows(skms, skcnt, skfl, skby),
                "kernels_note": "main-stream launches run back to back and own the chip: their event intervals are kernel times; the "
                                "few-row launches on the lanes' side streams run UNDERNEATH them (their intervals overlap those and "
                                "each other: time_share is event time / wall time, not a share of the critical path)",
                "end_to_end": {"achieved": e2e, "peak": PEAK_FP32_EQUIV_TFLOPS, "unit": "TFLOP/s", "frac": e2e / PEAK_FP32_EQUIV_TFLOPS,
                               "note": "ALL algorithmic fp32 FLOPs of the timed region (both MFMA classes on every stream + the folded map "
                                       "encoder) / wall time of the timed region / the split-operand roof"}}
        cpu = None
        if not args.no_cpu_baseline:                          # rank 0's host cores, whatever the world size
            cpu = cpu_baseline(cfg, w, scns, args)
        m, _ = acc.compute()
        sizes = sorted({cuts[i + 1] - cuts[i] for i in range(K)})
        shape = (N, R, args.polylines)
        cfg_tag = ("configs[4] (reward-tilt sweep: 8 tilt values x 1024 scenarios)" if (args.tilt_sweep and S * world == 8192 and shape == (64, 90, 512)) else
                   "configs[3] (8192 scenarios sharded over the ranks)" if (world > 1 and S * world == 8192 and shape == (64, 90, 512)) else
                   "configs[2]" + (" per GPU" if world > 1 else "") if (S,) + shape == (2048, 64, 90, 512) else
                   "configs[1]" if (S,) + shape == (256, 32, 90, 200) else "custom shape")
        if args.context_slots or args.context_polylines:      # reported separately, flagged non-reference
            cfg_tag = f"NON-REFERENCE secondary (wide context: A={d.A} slots, P={d.P} polylines per context; the reference's cfg has 24 / 200) on " + cfg_tag
        out = {
            "metric": "agent-steps/sec (closed-loop rollout), 64 agents x 90 steps",
            "dtype_short": f"f32 ({'f16x3' if f16 else 'bf16x6'} split-operand MFMA, fp32 accumulate)",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": f"f32 via {'f16x3' if f16 else 'bf16x6'} split operands ({scheme} 16-bit MFMA products per fp32 product, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload_tag": f"BASELINE.json {cfg_tag}: {S} synthetic scenarios/GPU x {N} agents x {R} steps, {args.polylines} polylines, "
                                       f"CtRL-Sim base model (random init)" + (", tilt sweep" if args.tilt_sweep else ""),
                       "workload": f"{S} synthetic Waymo-shaped scenarios resident per GPU x {N} agents x {R} steps, "
                                   f"{args.polylines} road polylines x 100 points, CtRL-Sim base model (8.29M params, "
                                   f"random init), context A={d.A}/T={d.T}/P={d.P}"
                                   + (" = BASELINE.json configs[4] (reward-tilt sweep: 8 tilt values x 1024 scenarios)"
                                      if (args.tilt_sweep and S * world == 8192 and (N, R, args.polylines) == (64, 90, 512)) else
                                      " = BASELINE.json configs[3] (8192 scenarios sharded over the ranks)"
                                      if (world > 1 and S * world == 8192 and (N, R, args.polylines) == (64, 90, 512)) else
                                      " = BASELINE.json configs[2]" + (" per GPU" if world > 1 else "")
                                      if (S, N, R, args.polylines) == (2048, 64, 90, 512) else
                                      " = BASELINE.json configs[1]" if (S, N, R, args.polylines) == (256, 32, 90, 200) else "")
                                   + f"; one bench step = the {R}-step closed-loop rollout of one slice of {'/'.join(map(str, sizes))} "
                                     f"scenarios, the {K} timed steps cover the {S} scenarios exactly once",
                       "scenarios_per_gpu": S, "scenarios_per_step": sizes, "agents": N, "rollout_steps": R,
                       "polylines": args.polylines, "model_batch_contexts": args.max_ctx, "lanes": args.lanes,
                       "lanes_pipelined": bool(args.pipeline),
                       "phases": None if args.pipeline else {"cached_s": cached_s, "sliding_s": sliding_s,
                                  "note": "main-stream time of the timed rollouts until the last lane of a slice left its K/V-cached steps "
                                          "(t < 32: few-row kernels only, on the side streams) / after it (full recompute per step)"},
                       "model_batch_reduced": any(c != max_ctx_asked for c in rank_ctx), "model_batch_contexts_requested": max_ctx_asked,
                       "model_batch_contexts_per_rank": rank_ctx,
                       "rank_elapsed_s": rank_times,
                       "contexts_per_rollout_rank0": ctx_per_rollout,
                       "mean_focal_groups_per_scenario_step": ctx_per_rollout / (S * R),
                       "scenario_upload_ms_untimed": upload_ms, "scenario_generation_s_untimed": gen_s,
                       "tilt": "sweep of 8 values, one per scenario (configs[4])" if args.tilt_sweep else list(args.tilt),
                       "size_classes": list(eng.sizes),
                       "few_row_kernels_on_side_streams": {"second_pass": eng.pass2_on_side, "first_pass_tail": eng.tail_on_side,
                                                           "cached_steps": eng.cached_on_side},
                       "collective": (f"one all-reduce (SUM) of the {n_vec}-double metric vector + barriers, backend {backend}, world {world}"),
                       "parallelism": f"scenario-sharded x{world}"},
            "roofline": roof, "cpu_baseline": cpu, "parity_spot_check": spot,
            "clock_power_during_timed_region": clock_samples,
            "rollout_metrics": {k: (None if v != v else v) for k, v in m.items()},
        }
        if args.fallback_slice and f16:
            out["fallback_bf16x6"] = _fallback_price(args, cfg, w, device, tilt, scns, cuts, R, N, value, eng)
        _emit(out, args.detail_file)


class _ClockSampler:
    """Shader clock and socket power of rank 0's GPU while the timed region runs (round 6: run alone, every hot kernel drives the socket to
    its 1.4 kW cap and the clock settles where the budget allows — 1.57 to 2.33 GHz, profiles/r06_power_clocks.md; the rollout averages
    1.2 kW at 2.1-2.3 GHz — so the clock a run sustained belongs in its record).  A host thread calls `rocm-smi --showclocks --showpower -d <gpu>` every few seconds (sysfs reads: nothing is queued on the
    GPU); no rocm-smi, no samples."""

    def __init__(self, gpu, period_s=4.0):
        import shutil
        import threading
        self.gpu, self.period, self.samples = int(gpu), period_s, []
        self.exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True) if self.exe else None
        if self._th:
            self._th.start()

    def _run(self):
        import re
        import subprocess
        while not self._stop.wait(self.period):
            try:
                out = subprocess.run([self.exe, "--showclocks", "--showpower", "-d", str(self.gpu)], capture_output=True, text=True, timeout=10).stdout
            except (OSError, subprocess.SubprocessError):
                continue
            m = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", out)
            p = re.search(r"Power \(W\):\s*([\d.]+)", out)
            if m and p:
                self.samples.append((int(m.group(1)), float(p.group(1))))

    def stop(self):
        if not self._th:
            return None
        self._stop.set()
        self._th.join(timeout=15)
        if not self.samples:
            return None
        clk = sorted(s[0] for s in self.samples)
        pw = sorted(s[1] for s in self.samples)
        return {"samples": len(self.samples), "sclk_mhz_median": clk[len(clk) // 2], "sclk_mhz_min": clk[0], "sclk_mhz_max": clk[-1],
                "socket_power_w_median": pw[len(pw) // 2], "socket_power_w_max": pw[-1], "source": "rocm-smi --showclocks --showpower, every 4 s"}


def _self_launch(n):
    """Run `sys.argv` as n ranks of one node (torch.distributed.run, rendezvous on 127.0.0.1 at a free port) and pass the exit code on."""
    import socket
    import subprocess
    if "--dry-run" not in sys.argv and torch.cuda.is_available() and torch.cuda.device_count() < n \
            and os.environ.get("CTRLSIM_BENCH_DEBUG_SHARED_GPU") != "1":
        raise SystemExit(f"bench.py: --gpus {n} but this node shows {torch.cuda.device_count()} GPU(s)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


SHORT_LINE_LIMIT = 4096          # bytes: the driver parses the LAST stdout line; round 5's 27 KB line came back as parsed: null


def short_line(out, detail_path):
    """The ONE stdout line of a run: the contract's fields + roofline of the dominant KERNEL + cpu_baseline, well under SHORT_LINE_LIMIT.
    Everything else (per-kernel rows of both streams, per-class attention rows, satellites, notes) lives in the detail file."""
    r = out.get("roofline") or {}
    rows = r.get("kernels") or []
    dom = rows[0] if rows else {}                                  # kernel_rows() sorts by time share: the dominant kernel of the run
    att = next((k for k in rows if k.get("kind") == "attention_causal"), {})
    c = out["config"]
    cpu = out.get("cpu_baseline")
    rnd = lambda v, n=4: None if v is None else float(f"{v:.{n}g}")
    line = {
        "metric": out["metric"], "value": rnd(out["value"], 7), "unit": out["unit"], "n_gpus": out["n_gpus"], "steps": out["steps"],
        "warmup": out["warmup"], "ms_per_step": rnd(out["ms_per_step"], 7), "higher_is_better": True, "scaling": out["scaling"],
        "vs_baseline": out["vs_baseline"], "dtype": out["dtype_short"], "data": out["data"],
        "config": {"workload": c["workload_tag"], "scenarios_per_gpu": c["scenarios_per_gpu"], "agents": c["agents"],
                   "rollout_steps": c["rollout_steps"], "polylines": c["polylines"], "model_batch_contexts": c["model_batch_contexts"],
                   "lanes": c["lanes"], "parallelism": c["parallelism"]},
        "roofline": {"bound": r.get("bound"), "kernel": dom.get("kind"), "achieved": rnd(dom.get("achieved")), "peak": rnd(r.get("peak")),
                     "unit": r.get("unit"), "frac": rnd(dom.get("frac")), "traffic": rnd(dom.get("traffic")),
                     "algorithmic_hbm_bytes_per_launch": rnd(dom.get("algorithmic_hbm_bytes_per_launch")),
                     "avg_launch_ms": rnd(dom.get("avg_launch_ms")), "launches": dom.get("launches"),
                     "time_share_of_step": rnd(dom.get("time_share_of_step")),
                     "linear_class_frac": rnd(r.get("frac")), "attention_causal_frac": rnd(att.get("frac")),
                     "end_to_end_frac": rnd((r.get("end_to_end") or {}).get("frac")),
                     "end_to_end_achieved": rnd((r.get("end_to_end") or {}).get("achieved"))},
        "cpu_baseline": None if not cpu else {"value": rnd(cpu["value"]), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                               "sample": cpu["sample_short"]},
        "parity_spot_check": None if not out.get("parity_spot_check") else {"identical": out["parity_spot_check"]["identical"]},
        "detail_file": detail_path,
    }
    if out.get("fallback_bf16x6"):
        line["fallback_bf16x6_value"] = rnd(out["fallback_bf16x6"].get("value"))
    cs = out.get("clock_power_during_timed_region")
    if cs:
        line["sclk_mhz"], line["socket_power_w"] = cs["sclk_mhz_median"], cs["socket_power_w_median"]
    return line


def _emit(out, detail_path):
    """Detail -> file + stderr (first), then the short line alone on stdout (last)."""
    blob = json.dumps(out)
    rel = detail_path
    try:
        with open(detail_path, "w") as f:
            f.write(blob + "\n")
        rel = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT) else detail_path
    except OSError as e:                                           # a read-only tree must not cost the run its line
        print(f"[bench] detail file not written: {e}", file=sys.stderr)
        rel = None
    print("[bench detail] " + blob, file=sys.stderr)
    sys.stderr.flush()
    # whatever native libraries still hold in C stdio buffers (RCCL prints its version banner to stdout through one, and it would
    # otherwise be flushed at process exit, i.e. AFTER the line) goes out first: the JSON line must be the LAST line of stdout
    try:
        C.CDLL(None).fflush(None)
    except OSError:
        pass
    line = json.dumps(short_line(out, rel))
    if len(line) >= SHORT_LINE_LIMIT:
        raise RuntimeError(f"bench line is {len(line)} bytes: the driver's parser needs < {SHORT_LINE_LIMIT}")
    print(line)
    sys.stdout.flush()


def _fallback_price(args, cfg, w, device, tilt, scns, cuts, R, N, value, eng):
    """Untimed: the first slice once more on the bf16x6 operand split (three bf16 planes, six MFMA products per fp32 product) — what a
    checkpoint whose activations trip the fp16 range guard of the default split pays after the automatic fallback (engine.check_finite).
    Same engine, same workspace (sized for the three-plane K/V images when split = "auto"), same slice schedule; the weights' bf16 planes are
    resident already.  Leaves the engine on the split it found."""
    a, b = cuts[0], cuts[1]
    scheme0 = eng.scheme
    eng.scheme = 0
    eng._bind()
    try:
        eng.reset(a, b); eng.run(R, s0=a, s1=b)                   # warm-up of the bf16x6 code objects
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.reset(a, b); eng.run(R, s0=a, s1=b)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        eng._unchecked = eng._unchecked[:-2]
    finally:
        eng.scheme = scheme0
        eng._bind()
    v = (b - a) * N * R / el
    return {"value": v, "unit": "agent-steps/s (this rank)", "slice_scenarios": b - a, "seconds": el, "ratio_to_f16x3_per_rank": v / (value / max(1, int(os.environ.get("WORLD_SIZE", "1")))),
            "note": "one untimed slice rolled on the three-bf16-plane split (roof 2500 / 6 TFLOP/s fp32-equivalent) by the same engine"}


def _sustained(nprod, ms, fl, dom):
    """The matrix rate the part SUSTAINS (profiles/r05_mfma_sustained.json: a register-only MFMA loop on random data runs at 1.48 PFLOP/s —
    the chip clocks to its power budget, 1.42 GHz under a pure matrix load, where the data sheet's 2.5 PFLOP/s is 2.4 GHz) next to the
    data-sheet peak the `frac` fields are quoted against."""
    path = os.path.join(ROOT, "profiles", "r05_mfma_sustained.json")
    if not os.path.exists(path) or ms[dom] <= 0:
        return None
    m = json.load(open(path))
    peak = m["mfma_16bit_sustained_tflops"] / nprod
    a = fl[dom] / (ms[dom] * 1e-3) / 1e12
    return {"peak_fp32_equivalent": peak, "unit": "TFLOP/s", "frac_of_sustained": a / peak, "mfma_16bit_sustained_tflops": m["mfma_16bit_sustained_tflops"],
            "implied_clock_ghz": m["implied_clock_ghz"], "source": "profiles/r05_mfma_sustained.json",
            "note": "informational: `frac` above stays relative to the data-sheet peak (MI355X_MICROARCH.md)"}


def _reduce_times(dist, elapsed, t_own, max_ctx, coll_device, gather_scalar):
    """The timed region's wall time is the MAX over ranks (the contract); next to it every rank's OWN time to its last kernel — a straggler
    shows as max >> min in the driver's SCALE line — and every rank's model batch (a rank that ran out of device memory halves its own)."""
    own = gather_scalar(dist, t_own, coll_device)
    ctx = [int(round(v)) for v in gather_scalar(dist, float(max_ctx), coll_device)]
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
    if dist is not None:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    return float(t_el.item()), {"min": min(own), "max": max(own), "per_rank": own,
                                "note": "each rank's own time from the start barrier to the end of its last rollout (before it waits at the "
                                        "closing barrier); value is computed from the MAX over ranks of the barrier-to-barrier time"}, ctx


class _DryEngine:
    """--dry-run only: stands where RolloutEngine stands in main() so that the rank flow around it is the real one.  It rolls nothing — a
    run() sleeps a time proportional to the scenarios of the slice (rank 3 a little longer: a straggler) — and its metric vector is the
    packed accumulators of FAKE rollouts keyed by the GLOBAL scenario ids, so the all-reduced vector must equal a single process's over
    the union of the ids whatever the world size.  Environment: CTRLSIM_BENCH_DRY_OOM_RANKS="3,5" makes those ranks fail their first
    engine construction with torch.OutOfMemoryError, as a GPU with less free memory would."""
    _failed_once = False

    def __init__(self, cfg, rank, ids, max_ctx):
        oom = {int(x) for x in filter(None, os.environ.get("CTRLSIM_BENCH_DRY_OOM_RANKS", "").split(","))}
        if rank in oom and not _DryEngine._failed_once:
            _DryEngine._failed_once = True
            raise torch.OutOfMemoryError("dry run: simulated out of device memory")
        self.cfg, self.rank, self.ids, self.max_ctx = cfg, rank, list(ids), max_ctx
        self.sizes, self.record_phases, self.phase_events, self.full_pass_contexts = (24,), False, [], 0
        self.rolled = []

    def load_scenarios(self, scns, steps=None):
        self.scns = scns

    def reset(self, a, b):
        pass

    def run(self, steps, s0=0, s1=None):
        time.sleep(0.002 * (s1 - s0) * (1.5 if self.rank == 3 else 1.0))
        self.rolled.append((s0, s1))

    def run_jobs(self, jobs, steps):
        for a, b in jobs:
            self.run(steps, a, b)

    def phase_times(self):
        return 0.0, 0.0

    def fake_metric_vector(self, metrics):
        acc = metrics.MetricAccumulators()
        T1 = self.cfg.nocturne.steps + 1
        for gid, scn in zip(self.ids, self.scns):
            rs = np.random.RandomState(1000 + gid)
            N = scn.N
            st = np.zeros((N, T1, 8))
            st[..., :2] = np.stack([scn.x, scn.y], 1)[:, None] + np.cumsum(rs.normal(0, 0.5, (N, T1, 2)), 1)
            st[..., 2:4] = rs.normal(0, 3, (N, T1, 2)); st[..., 4] = rs.uniform(-3, 3, (N, T1)); st[..., 7] = 1
            coll = (rs.uniform(size=(N, T1, 2)) < 0.01).astype(np.uint8)
            gt = np.zeros((N, T1, 5)); gt[..., :2] = st[..., :2] + rs.normal(0, 1, (N, T1, 2)); gt[..., 3] = 5; gt[..., 4] = 1
            acc.add_scenario(st, coll, rs.uniform(-10, 10, (N, T1)), gt, scn.goal_pos.astype(float), scn.goal_heading.astype(float),
                             scn.goal_speed.astype(float), self.cfg)
        return torch.from_numpy(acc.pack())


def _finish_dry(args, dist, rank, world, eng, cfg, scns, ids, elapsed, t_own, max_ctx_asked, coll_device, gather_scalar, metrics, tilt):
    """The tail of main() for --dry-run: the same collectives in the same order as the real run (time reduction, SUM all-reduce of the
    metric vector, parting barrier, destroy), then rank 0 alone takes its CPU sample (here: a sleep) and prints."""
    elapsed, rank_times, rank_ctx = _reduce_times(dist, elapsed, t_own, args.max_ctx, coll_device, gather_scalar)
    vec = eng.fake_metric_vector(metrics).to(coll_device)
    if dist is not None:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    backend = dist.get_backend() if dist is not None else "none (single process)"
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    time.sleep(float(os.environ.get("CTRLSIM_BENCH_DRY_CPU_S", "0")))     # rank 0's CPU baseline: the other ranks have left already
    S, N, R, K = args.scenarios, args.agents, args.rollout_steps, args.steps
    tl = np.asarray(tilt, np.float64)
    detail = {"dry_run": True, "metric": "DRY RUN of the rank flow — not a measurement", "value": None, "n_gpus": world, "steps": K,
              "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3, "scaling": "weak",
              "config": {"scenario_ids_rank0": ids, "tilt_rank0": tl[:, 0].tolist() if tl.ndim == 2 else tl.tolist(),
                         "model_batch_contexts_requested": max_ctx_asked, "model_batch_contexts_per_rank": rank_ctx,
                         "model_batch_reduced": any(c != max_ctx_asked for c in rank_ctx), "rank_elapsed_s": rank_times,
                         "collective": f"one all-reduce (SUM) of the {int(vec.numel())}-double metric vector + barriers, backend {backend}, world {world}"},
              "agent_steps_counted": S * N * R * world, "metric_vector": vec.cpu().numpy().tolist()}
    # the same split as a real run: the bulky part (here the 1250-double metric vector and the per-rank lists) goes to the detail file,
    # the stdout line stays under SHORT_LINE_LIMIT whatever the world size
    with open(args.detail_file, "w") as f:
        f.write(json.dumps(detail) + "\n")
    line = json.dumps({k: detail[k] for k in ("dry_run", "metric", "value", "n_gpus", "steps", "warmup", "ms_per_step", "scaling",
                                               "agent_steps_counted")} | {"detail_file": args.detail_file})
    assert len(line) < SHORT_LINE_LIMIT
    print(line)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, w, scns, args):
    """The CPU oracle (reference cost structure: two dense B=1 forwards per focal group and step, C physics) on the
    host cores of this box, on a bounded sample of the same workload: the first steps of several scenarios (the
    reference's cost per step does not depend on t: it always runs the dense T=32 forward)."""
    import rollout_oracle
    import sim_libs
    sim_libs.build_oracle()
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    import model_oracle
    model_oracle.FUSED_SDPA = True                     # attention through torch's fused op, as nn.MultiheadAttention evaluates it in the reference
    ro = rollout_oracle.RolloutOracle(cfg, w, seed=args.seed, threads=threads)
    k, ns = args.cpu_sample_steps, min(args.cpu_sample_scenarios, len(scns))
    groups = 0
    t0 = time.perf_counter()
    done = 0
    for scn in scns[:ns]:
        if done and time.perf_counter() - t0 > args.cpu_sample_budget_s:
            break                                                      # time box: the default run must finish within minutes
        o = ro.run(scn, k, sim_libs.OracleSim, dense_window=True)     # the reference forwards the full T = 32 window at every step
        groups += int(o["n_groups"].sum())
        done += 1
    ns = done
    el = time.perf_counter() - t0
    return {"value": ns * scns[0].N * k / el, "unit": "agent-steps/s", "cores": threads, "kind": "port",
            "cpu_model": cpu_model(), "host_logical_cpus": cores,
            "sample_short": f"{ns} scenarios x {scns[0].N} agents x {k} steps ({groups} focal-group steps, {el:.1f} s, oracle port, {threads} threads)",
            "sample": f"{ns} scenarios x {scns[0].N} agents x {k} rollout steps ({groups} focal-group steps, "
                      f"{el:.1f} s: two dense T = 32 forwards per focal-group step, as the reference); oracle/rollout_oracle.py + "
                      f"oracle/sim_oracle.c, torch {torch.__version__} CPU, {threads} threads",
            "note": "rounds 1-3 timed the port on its first steps with buffers of `steps` rows, i.e. 2-step windows instead of the reference's "
                    "32: their cpu_baseline values (21-32 agent-steps/s) were several times too fast; on the build container's 8 cores the port "
                    "and the UNMODIFIED reference take 0.86 s and 0.94 s per focal-group step (profiles/r04_cpu_port_vs_reference.txt)"}


if __name__ == "__main__":
    main()
