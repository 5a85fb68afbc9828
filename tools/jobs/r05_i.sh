#!/bin/bash
# Round 5, job I: where the batched evaluator route's GPU time goes at 256 scenes; new wide trained fixture tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sim_ctx.py -m gpu -x -q -k "trained" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
( cd /tmp && export TMPDIR=/tmp && cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $O/evalprof -o ev --output-format csv -- python tools/facade_rate.py 255 8 90 batched > $O/evalprof.log 2>&1 )
find $O/evalprof -name "*_kernel_trace.csv" -delete
tail -2 $O/evalprof.log
