#!/bin/bash
# Round 6, job S: cfg.model.attend_own_return_action (mask mode 5, dims.variant 4) — its new tests first, then the full GPU suite and a short bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_s; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "own_return" > $O/own_ops.txt 2>&1; tail -15 $O/own_ops.txt
timeout 900 python -m pytest tests/test_gpu_sim_ctx.py -q -m gpu -k "own_return" > $O/own_model.txt 2>&1; tail -40 $O/own_model.txt
timeout 2400 python -m pytest tests -q -m gpu -x > $O/suite.txt 2>&1; tail -5 $O/suite.txt
timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile --fallback-slice 0 --detail-file $O/d.json > $O/b.json 2> $O/bench_err.txt; tail -1 $O/b.json | cut -c1-600
