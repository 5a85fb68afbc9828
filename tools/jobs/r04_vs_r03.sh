#!/bin/bash
# Same-box comparison of the round-3 tree (scratch/r03tree: git worktree of 26d3fcc, built there) with the current tree.
O=$PWD/gpurun_out/r04_vs_r03; mkdir -p $O
C1="--scenarios 256 --agents 32 --polylines 200 --no-cpu-baseline --spot-check 0"
C2="--scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 0"
for rep in 1 2; do
  (cd scratch/r03tree && timeout 600 python bench.py $C1 > $O/r03_c1_$rep.json 2> $O/r03_c1_$rep.err)
  timeout 600 python bench.py $C1 --no-class-profile > $O/r04_c1_$rep.json 2> $O/r04_c1_$rep.err
  (cd scratch/r03tree && timeout 600 python bench.py $C2 > $O/r03_c2_$rep.json 2> $O/r03_c2_$rep.err)
  timeout 600 python bench.py $C2 --no-class-profile > $O/r04_c2_$rep.json 2> $O/r04_c2_$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_vs_r03/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, round(d["value"]), "ms/step", round(d["ms_per_step"],1))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-300:])
PY
