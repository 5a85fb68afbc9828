#!/bin/bash
# Round 6, job Z: wide context (A = 64, P = 512; SURVEY 8(d) secondary, non-reference) — closed loop against the oracle, then its rate
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_z; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sim_ctx.py -q -m gpu -k "wide_context" -x 2>&1 | tail -30 | tee $O/wide_loop.txt
timeout 900 python bench.py --scenarios 204 --steps 2 --warmup 1 --context-slots 64 --context-polylines 512 --no-cpu-baseline --spot-check 2 --no-class-profile --fallback-slice 0 --detail-file $O/d_wide.json > $O/b_wide.json 2> $O/bench_err.txt
tail -1 $O/b_wide.json | cut -c1-1500; tail -5 $O/bench_err.txt | cut -c1-300
