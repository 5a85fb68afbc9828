#!/bin/bash
# Round 5, job AC: weight-stationary Linear without K / V images: results from registers, ONE barrier per 32-row block
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_ac; mkdir -p $O
cd $R
for v in new H5; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_$v.so; fi
  echo "$v: $(CTRLSIM_LIB=$L timeout 300 python tools/microbench/ws256_bits.py 2>&1 | tail -1)" | tee -a $O/bits.txt
done
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $O/pytest_op.txt 2>&1; tail -3 $O/pytest_op.txt
for v in new H5 new H5; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_$v.so; fi
  echo "$v: $(CTRLSIM_LIB=$L timeout 300 python tools/microbench/fusion_proxies.py 2>&1 | grep 'ws256' | tr '\n' ' ')" | tee -a $O/timing.txt
done
for v in new H5 new H5 new H5; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_$v.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],1), d["parity_spot_check"]["identical"])
PY
done
