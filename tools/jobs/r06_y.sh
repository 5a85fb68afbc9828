#!/bin/bash
# Round 6, job Y: the wide context (A = 64, P = 512, SURVEY 8(d) secondary) through the HIP forward against the oracle
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_y; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "forward_matches_oracle and wide" -x 2>&1 | tail -30 | tee $O/wide.txt
