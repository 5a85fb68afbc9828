#!/bin/bash
# Round 5, job T: option 3 = 2 (out-projection fused into the feed-forward kernel only) against 3 (+ self out-projection / norm1 / cross-Q kernel), final build, interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_t; mkdir -p $O
cd $R
for v in 2 3 2 3 2 3 1; do
  CTRLSIM_OPTIONS=3=$v timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 0 --no-class-profile > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("option 3 =", sys.argv[2], round(d["value"]), round(d["ms_per_step"],1))
PY
done
