#!/bin/bash
# Round 6, job I: key-padded attention with the (context, head)'s keys resident in LDS (RES) against the staged kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
for v in new _DATT_KEYPAD_RES_0 new _DATT_KEYPAD_RES_0; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  echo "== $v" | tee -a $O/sustained.txt
  CTRLSIM_LIB=$L timeout 600 python tools/microbench/sustained.py 256 1.0 "attn" 2>&1 | grep -E "^attn (cross)" | tee -a $O/sustained.txt
done
for v in new _DATT_KEYPAD_RES_0 new _DATT_KEYPAD_RES_0; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 0 --no-class-profile --fallback-slice 0 --detail-file $O/d_$v.json > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $O/d_$v.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); dd=json.load(open(sys.argv[2]))
kp=[k for k in dd["roofline"]["kernels"] if k["kind"]=="attention_keypad"]
print(sys.argv[3], round(d["value"]), round(d["ms_per_step"],1), "keypad frac", [round(k["frac"],3) for k in kp], "avg ms", [round(k["avg_launch_ms"],3) for k in kp])
PY
done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_sim_ctx.py -m gpu -x -q > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
