#!/bin/bash
# Round 5, job Y: stamps of the PRE mode as shipped now; model / closed-loop fixtures on this build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_y; mkdir -p $O
cd $R
CTRLSIM_LIB=$R/tools/microbench/variants/all_preS.so timeout 200 python tools/microbench/pre_stamps.py 2>&1 | tail -8 | tee $O/pre_stamps.txt
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_sim_ctx.py -m gpu -x -q > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
