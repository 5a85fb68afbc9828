#!/bin/bash
# Round 6, job E: two 32-query groups per wave (QG = 2) in the two full-row attention kernels: tests, sustained, bench A/B against the control-word
# kernel (ctl: round 6 before QG) and round 5's control flow (r5flow); map_pool with its thread-per-point phase on all four waves
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_e; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
for v in new ctl r5flow; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  echo "== $v" | tee -a $O/sustained.txt
  SUSTAINED_CLASSES=5,8,12,16,20 CTRLSIM_LIB=$L timeout 600 python tools/microbench/sustained.py 256 1.0 "attn compact" 2>&1 | grep -E "^attn (causal mask|cross|mask-table)" | tee -a $O/sustained.txt
done
CTRLSIM_LIB=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so timeout 300 python tools/microbench/map_pool.py 1024 2>&1 | tail -3 | tee $O/map_pool_new.txt
CTRLSIM_LIB=$R/tools/microbench/variants/mapold.so timeout 300 python tools/microbench/map_pool.py 1024 2>&1 | tail -3 | tee $O/map_pool_old.txt
for v in new ctl new ctl r5flow; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 0 --no-class-profile --fallback-slice 0 --detail-file $O/d_$v.json > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],1), d["roofline"]["attention_causal_frac"])
PY
done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_sim_ctx.py -m gpu -x -q > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
