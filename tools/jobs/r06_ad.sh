#!/bin/bash
# Round 6, job AD: wave priorities in the attention kernels (-DATT_PRIO=1: younger half of the 8-wave workgroups at priority 1; =2: softmax section at priority 1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ad; mkdir -p $O
cd $R
for v in base _DATT_PRIO_1 _DATT_PRIO_2; do
  if [ $v = base ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  echo "== $v" | tee -a $O/sustained.txt
  SUSTAINED_CLASSES=5,8,12,16,20 CTRLSIM_LIB=$L timeout 600 python tools/microbench/sustained.py 256 0.8 "attn compact" 2>&1 | grep -E "^attn (causal mask|mask-table|cross)" | tee -a $O/sustained.txt
done
for v in base _DATT_PRIO_1 _DATT_PRIO_2 base _DATT_PRIO_1 _DATT_PRIO_2; do
  if [ $v = base ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile --fallback-slice 0 --detail-file $O/d_$v.json > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],1), d["roofline"]["attention_causal_frac"], d.get("sclk_mhz"), d.get("socket_power_w"), d["parity_spot_check"])
PY
done
