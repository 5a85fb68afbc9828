#!/bin/bash
# Generic same-box A/B of ctrlsim_set_option switches: r04_ab.sh NAME "OPTS_A" "OPTS_B" [pytest -k expression]
NAME=$1; OA=$2; OB=$3; K=$4
O=gpurun_out/r04_ab_$NAME; mkdir -p $O
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt; fi
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile"
for rep in 1 2; do
  CTRLSIM_OPTIONS=$OA timeout 600 $B > $O/a_$rep.json 2> $O/a_$rep.err
  CTRLSIM_OPTIONS=$OB timeout 600 $B > $O/b_$rep.json 2> $O/b_$rep.err
done
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, round(d["value"]), "spot", d["parity_spot_check"]["identical"], "linear frac", round(d["roofline"]["frac"],4), "e2e", round(d["roofline"]["end_to_end"]["frac"],4),
              [(r["kernel"][:22], r["launches"], round(r["avg_launch_ms"],4)) for r in d["roofline"]["kernels"][:6]])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
