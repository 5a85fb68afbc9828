#!/bin/bash
# Round 5, job M: QP mode with both outputs leaving from the accumulator layout underneath the products
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_m; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -s -k "outproj or ffn_fused" > $O/pytest_op.txt 2>&1; grep -a "passed\|failed" $O/pytest_op.txt | tail -3
timeout 300 python tools/microbench/fusion_proxies.py > $O/fusion_proxies.txt 2>&1; tail -8 $O/fusion_proxies.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_sim_ctx.py -m gpu -x -q > $O/pytest_model.txt 2>&1; tail -3 $O/pytest_model.txt
for v in 3 2 3; do
  CTRLSIM_OPTIONS=3=$v timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile > $O/bench_opt3_$v.$RANDOM.json 2> $O/bench_err.txt
done
for f in $O/bench_opt3_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"])
PY
done
