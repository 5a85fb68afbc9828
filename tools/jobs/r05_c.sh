#!/bin/bash
# Round 5, job C: attention per-segment timing, per-kernel PMC traffic of the bench command, full GPU suite (trained-like parity, configs[1] at 256 scenes, guard pair, options)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_c; mkdir -p $O
cd $R
timeout 300 tools/microbench/att_timing2 64 > $O/att_timing.txt 2>&1; cat $O/att_timing.txt
timeout 1200 bash tools/pmc_traffic.sh $O/pmc -- python bench.py --scenarios 102 --steps 1 --warmup 0 --no-cpu-baseline --spot-check 0 --no-class-profile 2>&1 | tail -12
find $O/pmc -name "*counter_collection.csv" -size +8M -delete
timeout 1700 python -m pytest tests -m gpu -x -q -s > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt; grep -E "ours vs reference|at-scale" $O/pytest.txt | head -20
