#!/bin/bash
# Round 5, job D: causal / cross attention with 256 queries per workgroup (-DATT_NW=8) against the shipped 128; facade tests (batched evaluator route)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_d; mkdir -p $O
cd $R
for v in base _DATT_NW_8; do
  echo "== $v" | tee -a $O/attn_variants.txt
  CTRLSIM_LIB=$R/tools/microbench/variants/$v.so SUSTAINED_CLASSES=5,8,12,16,20 timeout 600 python tools/microbench/sustained.py 256 1.2 attn+compact 2>&1 | grep -E "attn" | tee -a $O/attn_variants.txt
done
CTRLSIM_LIB=$R/tools/microbench/variants/_DATT_NW_8.so timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "attention or forward or compact" > $O/pytest_nw8.txt 2>&1; tail -3 $O/pytest_nw8.txt
timeout 1200 python -m pytest tests/test_gpu_facade.py -m gpu -x -q -s > $O/pytest_facade.txt 2>&1; tail -5 $O/pytest_facade.txt; grep "per-scenario route" $O/pytest_facade.txt
timeout 300 python tools/facade_rate.py 15 8 20 2>&1 | tail -1 | tee $O/facade_rate.txt
timeout 300 python tools/facade_rate.py 63 8 20 batched 2>&1 | tail -1 | tee -a $O/facade_rate.txt
