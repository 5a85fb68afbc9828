#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_q; mkdir -p $O
cd $R
for v in S S1 S2 S3; do echo "== $v"; CTRLSIM_LIB=$R/tools/microbench/variants/all_qp$v.so timeout 200 python tools/microbench/qp_stamps.py 2>&1 | tail -7; done | tee $O/qp_stamps.txt
