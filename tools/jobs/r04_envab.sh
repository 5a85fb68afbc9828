#!/bin/bash
# Same-box A/B of an environment switch: r04_envab.sh NAME "ENV_A" "ENV_B" [pytest -k]
NAME=$1; EA=$2; EB=$3; K=$4
O=gpurun_out/r04_envab_$NAME; mkdir -p $O
if [ -n "$K" ]; then env $EB timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt; fi
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile"
for rep in 1 2; do
  env $EA timeout 600 $B > $O/a_$rep.json 2> $O/a_$rep.err
  env $EB timeout 600 $B > $O/b_$rep.json 2> $O/b_$rep.err
done
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        sat=d["roofline"]["satellite"]
        print(f, round(d["value"]), "spot", d["parity_spot_check"]["identical"], "e2e", round(d["roofline"]["end_to_end"]["frac"],4),
              "build_context ms", round(sat["build_context"]["avg_launch_ms"],3), "share", round(sat["build_context"]["time_share_of_step"],4))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
