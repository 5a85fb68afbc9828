#!/bin/bash
# Round 5, job G: full GPU suite on the attention build with 8-wave workgroups; where the batched evaluator route's GPU time goes; the round's profile set
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_g; mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
( cd /tmp && export TMPDIR=/tmp && cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $O/evalprof -o ev --output-format csv -- python tools/facade_rate.py 63 8 90 batched > $O/evalprof.log 2>&1 )
find $O/evalprof -name "*_kernel_trace.csv" -delete
f=$(find $O/evalprof -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-200; tail -3 $O/evalprof.log
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
