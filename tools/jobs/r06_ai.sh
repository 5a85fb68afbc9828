#!/bin/bash
# Round 6, job AI: the combined switch case (own-return mask + encode_initial_state False) against the reference fixture; clocks per kernel on the final build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ai; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sim_ctx.py -q -m gpu -k "model_flags" 2>&1 | tail -5 | tee $O/flags.txt
timeout 600 python tools/microbench/clock_by_kernel.py 256 6 2>&1 | grep -E "ms " | tee $O/clock_by_kernel.txt
