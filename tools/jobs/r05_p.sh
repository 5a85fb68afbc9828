#!/bin/bash
# Round 5, job P: residual rows converted per out-block underneath the leading product; QP prefetches the next row block's rows
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_p; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "outproj or ffn_fused" > $O/pytest_op.txt 2>&1; grep -a "passed\|failed" $O/pytest_op.txt | tail -3
for v in new H new H; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_qp$v.so; fi
  echo "$v: $(CTRLSIM_LIB=$L timeout 120 python tools/microbench/qp_timing.py 2>&1 | tail -1)" | tee -a $O/qp_timing.txt
done
timeout 300 python tools/microbench/fusion_proxies.py > $O/fusion_proxies.txt 2>&1; tail -4 $O/fusion_proxies.txt | head -2
for v in new H new H; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_qp$v.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile > $O/bench_$v.$RANDOM.json 2> $O/bench_err.txt
done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"])
PY
done
