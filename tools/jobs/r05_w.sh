#!/bin/bash
# Round 5, job W: closing LayerNorm of the PRE mode in the accumulator layout (no LDS-staged epilogue)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_w; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -s -k "outproj or ffn_fused" > $O/pytest_op.txt 2>&1; grep -a "one kernel: max\|passed\|failed" $O/pytest_op.txt | tail -8
for v in new H2; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_$v.so; fi
  echo "$v: $(CTRLSIM_LIB=$L timeout 300 python tools/microbench/fusion_proxies.py 2>&1 | grep 'BUILT (b)')" | tee -a $O/pre_timing.txt
done
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_sim_ctx.py -m gpu -x -q > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
for v in new H2 new H2 new H2; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_$v.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 0 --no-class-profile > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],1))
PY
done
