#!/bin/bash
# Round 6, job AF: three more rollout pairs for -DATT_DIRECT_STORE=1 (one run of job AE was 8 % slow with an unchanged attention rate)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_af; mkdir -p $O
cd $R
V=$R/tools/microbench/variants/_DATT_DIRECT_STORE_1.so
for v in ds base ds base ds base; do
  if [ $v = base ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$V; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile --fallback-slice 0 --detail-file $O/d_$v.json > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],1), d["roofline"]["attention_causal_frac"], d["roofline"]["frac"], d.get("sclk_mhz"), d.get("socket_power_w"), d["parity_spot_check"])
PY
done
