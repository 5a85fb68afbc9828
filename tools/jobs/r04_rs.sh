#!/bin/bash
# Row-stationary in_proj: GPU tests that touch it, sustained rows, same-box rollout A/B against the weight-stationary kernel (option 6 = 7).
O=gpurun_out/r04_rs; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/microbench/sustained.py 256 1.5 gemm 2>&1 | grep images | tee $O/sustained.txt
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile"
for rep in 1 2; do
  CTRLSIM_OPTIONS=6=7 timeout 600 $B > $O/a_$rep.json 2> $O/a_$rep.err
  timeout 600 $B > $O/b_$rep.json 2> $O/b_$rep.err
done
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, round(d["value"]), "spot", d["parity_spot_check"]["identical"], "e2e", round(d["roofline"]["end_to_end"]["frac"],4),
              [(r["kernel"][:14], round(r["avg_launch_ms"],4), round(r["frac"],3)) for r in d["roofline"]["kernels"][:6]])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
