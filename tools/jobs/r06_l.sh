#!/bin/bash
# Round 6, job L: the mask-table kernel's query blocks aligned to the END of the row range (partial block first) against the start-aligned blocks
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_l; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" > $O/pytest_ops.txt 2>&1; tail -2 $O/pytest_ops.txt
for v in new _DATT_ALIGN_END_0; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  echo "== $v" | tee -a $O/sustained.txt
  SUSTAINED_CLASSES=4,6,9,11,12,14,20 CTRLSIM_LIB=$L timeout 600 python tools/microbench/sustained.py 256 1.0 "compact" 2>&1 | grep -E "^attn (mask-table)" | tee -a $O/sustained.txt
done
for v in new _DATT_ALIGN_END_0 new _DATT_ALIGN_END_0; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile --fallback-slice 0 --detail-file $O/d_$v.json > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],1), d["roofline"]["attention_causal_frac"], d["parity_spot_check"])
PY
done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_sim_ctx.py -m gpu -x -q > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
