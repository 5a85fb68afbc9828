#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_ad; mkdir -p $O
cd $R
for v in new H5 new H5; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_$v.so; fi
  echo "$v: $(CTRLSIM_LIB=$L timeout 300 python tools/microbench/fusion_proxies.py 2>&1 | grep 'ws256' | tr '\n' ' ')" | tee -a $O/timing.txt
done
