#!/bin/bash
O=gpurun_out/r04_cfg1; mkdir -p $O
B="python bench.py --scenarios 256 --agents 32 --polylines 200 --no-cpu-baseline --no-class-profile --spot-check 0"
for rep in 1 2; do
for v in "base X=1" "no7 CTRLSIM_OPTIONS=7=0" "no8 CTRLSIM_OPTIONS=8=0" "no78 CTRLSIM_OPTIONS=7=0,8=0"; do
  set -- $v
  env $2 timeout 600 $B > $O/$1_$rep.json 2> $O/$1_$rep.err
  python - $O/$1_$rep.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(sys.argv[1], round(d["value"]), "ms/step", round(d["ms_per_step"],1), "phases", {k: round(v,3) for k,v in (d["config"]["phases"] or {}).items() if k!="note"})
PY
done; done
