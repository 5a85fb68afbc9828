#!/bin/bash
# Round 6, job AH: the shortened wide-context tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ah; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sim_ctx.py tests/test_gpu_model.py -q -m gpu -k "wide" --durations=3 2>&1 | tail -12 | tee $O/wide.txt
