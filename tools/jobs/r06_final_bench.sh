#!/bin/bash
# Round 6: the driver's default bench command on the final tree (stdout kept as the driver would see it) + wall time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_final; mkdir -p $O
cd $R
T0=$SECONDS
timeout 1200 python bench.py --detail-file $O/bench_detail.json > $O/bench_default.json 2> $O/bench_default.err
echo "wall $((SECONDS - T0)) s" | tee $O/wall.txt
wc -c $O/bench_default.json; tail -1 $O/bench_default.json
