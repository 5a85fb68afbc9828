#!/bin/bash
# Streaming form of the few-query attention launches (option 9): full GPU suite, same-box rollout A/B against the staged form.
O=gpurun_out/r04_dir; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile"
for rep in 1 2; do
  CTRLSIM_OPTIONS=9=0 timeout 600 $B > $O/a_$rep.json 2> $O/a_$rep.err
  timeout 600 $B > $O/b_$rep.json 2> $O/b_$rep.err
done
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, round(d["value"]), "spot", d["parity_spot_check"]["identical"], "e2e", round(d["roofline"]["end_to_end"]["frac"],4),
              [(r["kernel"][:14], round(r["avg_launch_ms"],4), round(r["frac"],3)) for r in d["roofline"]["kernels"][:6]],
              "side:", [(r["kernel"][:10], round(r["avg_launch_ms"],4), round(r["time_share_of_step"],3)) for r in d["roofline"]["kernels_on_side_streams"][:3]],
              "phases", {k: round(v, 2) for k, v in d["config"]["phases"].items() if k != "note"})
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
