#!/bin/bash
# Round 6, job Q: shader clock / socket power under each hot kernel; a rollout line with the clock sampler of bench.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_q; mkdir -p $O
cd $R
timeout 600 python tools/microbench/clock_by_kernel.py 256 6 > $O/clock_by_kernel.txt 2> $O/err.txt; cat $O/clock_by_kernel.txt; tail -3 $O/err.txt
timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 0 --no-class-profile --fallback-slice 0 --detail-file $O/d.json > $O/b.json 2> $O/bench_err.txt; cut -c1-200 $O/b.json; grep -o '"sclk_mhz.*' $O/b.json
