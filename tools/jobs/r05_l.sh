#!/bin/bash
# Round 5, job L: sustained times of the fused kernels, full GPU suite, driver-style bench + kernel trace with the fusions on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_l; mkdir -p $O
cd $R
timeout 300 python tools/microbench/fusion_proxies.py > $O/fusion_proxies.txt 2>&1; tail -14 $O/fusion_proxies.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 600 $O/bench_driver.json
( cd /tmp && export TMPDIR=/tmp && cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o r05t --output-format csv -- python bench.py --scenarios 204 --steps 2 --warmup 1 --spot-check 0 --no-class-profile --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err )
find $O/prof -name "*_kernel_trace.csv" -delete
find $O/prof -name "*kernel_stats.csv" | head -2
