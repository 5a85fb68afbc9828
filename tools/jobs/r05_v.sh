#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_v; mkdir -p $O
cd $R
CTRLSIM_LIB=$R/tools/microbench/variants/all_preS.so timeout 200 python tools/microbench/pre_stamps.py 2>&1 | tail -8 | tee $O/pre_stamps.txt
CTRLSIM_LIB=$R/tools/microbench/variants/all_preS.so timeout 200 python tools/microbench/pre_stamps.py 110592 2>&1 | tail -8 | tee -a $O/pre_stamps.txt
