#!/bin/bash
# Round 6, job W: few-row Linear launches — weight-stationary against tiled kernel per launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_w; mkdir -p $O
cd $R
timeout 600 python tools/microbench/few_rows.py 2>&1 | grep "M =" | tee $O/few_rows.txt
