#!/bin/bash
# Round 4: the cached-phase hole.  Same box: one run() per slice with two lanes starting together (rounds 1-3) against
# pipelined jobs with a third lane that rolls the next job's K/V-cached steps ahead of time (engine.run_jobs, at most two lanes in full-recompute steps).
O=gpurun_out/r04_lanes; mkdir -p $O
python ctrl-sim_amd/csrc/build.py > /dev/null 2>&1
timeout 600 python -m pytest tests/test_gpu_sim_ctx.py -x -q -k "pipelined_jobs" 2>&1 | tail -3
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2"
run() { name=$1; shift; timeout 900 $B "$@" > $O/$name.json 2> $O/$name.err; python - "$O/$name.json" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f, round(d["value"]), "spot", d["parity_spot_check"]["identical"], "max_ctx", d["config"]["model_batch_contexts"], "lanes", d["config"]["lanes"],
          "phases", {k: round(v, 2) for k, v in (d["config"].get("phases") or {}).items() if k != "note"}, "e2e", round(d["roofline"]["end_to_end"]["frac"], 4))
except Exception as e:
    print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
}
for rep in 1 2; do
  run base2_1024_$rep --lanes 2 --no-pipeline --max-ctx 1024
  run pipe2_1024_$rep --lanes 2 --max-ctx 1024
  run pipe3_768_$rep --lanes 3 --max-ctx 768
  run pipe4_640_$rep --lanes 4 --max-ctx 640
done
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04_lanes/pipe3_768_1.json") if l.startswith("{")][0])
for r in d["roofline"]["causal_attention_by_size_class"] or []:
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
PY
