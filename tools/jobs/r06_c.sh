#!/bin/bash
# Round 6, job C: three-stage K/V ring of the full-row attention kernels (default) against the two-stage ring and the no-DMA ablation
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention or scattered or linear256" > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
for v in new _DATT_NBUF_FULL_2 _DATT_NBUF_FULL_2_DATT_ABL_NODMA _DATT_ABL_NODMA; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  echo "== $v" | tee -a $O/sustained.txt
  SUSTAINED_CLASSES=5,8,12,16,20 CTRLSIM_LIB=$L timeout 600 python tools/microbench/sustained.py 256 1.0 "attn compact" 2>&1 | grep -E "^attn (causal mask|cross|mask-table)" | tee -a $O/sustained.txt
done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_sim_ctx.py -m gpu -x -q > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
