#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_aa; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "outproj" 2>&1 | tail -5 | tee $O/pytest_op.txt
