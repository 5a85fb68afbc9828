#!/bin/bash
O=gpurun_out/r04_lanes4; mkdir -p $O
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 0 --no-class-profile"
for cfg in "2 1024" "4 512" "3 640" "2 512"; do
  set -- $cfg
  timeout 900 $B --lanes $1 --max-ctx $2 > $O/l$1_$2.json 2> $O/l$1_$2.err
  python - $O/l$1_$2.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(sys.argv[1], round(d["value"]), "lanes", d["config"]["lanes"], "max_ctx", d["config"]["model_batch_contexts"], {k: round(v,3) for k,v in d["config"]["phases"].items() if k!="note"})
PY
done
