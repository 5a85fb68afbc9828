#!/bin/bash
# Round 6, job AE: attention output rows stored straight from the accumulators in every staged kernel (-DATT_DIRECT_STORE=1) against the LDS-transposed 4-byte stores
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ae; mkdir -p $O
cd $R
V=$R/tools/microbench/variants/_DATT_DIRECT_STORE_1.so
CTRLSIM_LIB=$V timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "attention" 2>&1 | tail -3 | tee $O/ops.txt
for v in base ds; do
  if [ $v = base ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$V; fi
  echo "== $v" | tee -a $O/sustained.txt
  SUSTAINED_CLASSES=5,8,12,16,20 CTRLSIM_LIB=$L timeout 600 python tools/microbench/sustained.py 256 0.8 "attn compact" 2>&1 | grep -E "^attn (causal mask|mask-table|cross)" | tee -a $O/sustained.txt
done
for v in base ds base ds; do
  if [ $v = base ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$V; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile --fallback-slice 0 --detail-file $O/d_$v.json > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],1), d["roofline"]["attention_causal_frac"], d.get("sclk_mhz"), d.get("socket_power_w"), d["parity_spot_check"])
PY
done
