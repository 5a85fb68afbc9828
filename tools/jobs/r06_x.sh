#!/bin/bash
# Round 6, job X: fused feed-forward kernels with the W1-shaped products on two alternating accumulator chains (-DFF_2CH=1) against the shipped single chain
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_x; mkdir -p $O
cd $R
V=$R/tools/microbench/variants/_DFF_2CH_1.so
CTRLSIM_LIB=$V timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "ffn or fused or feed or proj" 2>&1 | tail -3 | tee $O/ops.txt
for v in base 2ch base 2ch; do
  if [ $v = base ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$V; fi
  CTRLSIM_LIB=$L timeout 300 python tools/microbench/sustained.py 256 1.0 ffn 2>&1 | grep -E "^ffn" | sed "s/^/$v /" | tee -a $O/sustained.txt
done
for v in base 2ch base 2ch; do
  if [ $v = base ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$V; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile --fallback-slice 0 --detail-file $O/d_$v.json > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],1), d["roofline"]["frac"], d.get("sclk_mhz"), d.get("socket_power_w"), d["parity_spot_check"])
PY
done
