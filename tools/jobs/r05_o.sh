#!/bin/bash
# Round 5, job O: PRE / QP modes with their row loads requested in one batch (hipcc had serialised them: 16 round trips per row block)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_o; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "outproj or ffn_fused" > $O/pytest_op.txt 2>&1; grep -a "passed\|failed" $O/pytest_op.txt | tail -3
for v in new A; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_qp$v.so; fi
  echo "$v: $(CTRLSIM_LIB=$L timeout 120 python tools/microbench/qp_timing.py 2>&1 | tail -1)" | tee -a $O/qp_timing.txt
done
timeout 300 python tools/microbench/fusion_proxies.py > $O/fusion_proxies.txt 2>&1; tail -8 $O/fusion_proxies.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_sim_ctx.py -m gpu -x -q > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
for v in new A new A; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_qp$v.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile > $O/bench_$v.$RANDOM.json 2> $O/bench_err.txt
done
CTRLSIM_OPTIONS=3=1 timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile > $O/bench_opt1.json 2> $O/bench_err.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"])
PY
done
