#!/bin/bash
# Same-box A/B of the shipped library against a variant library: r04_libab.sh NAME VARIANT [sustained filter] [pytest -k]
NAME=$1; V=$PWD/tools/microbench/variants/all_$2.so; F=$3; K=$4
O=gpurun_out/r04_libab_$NAME; mkdir -p $O
if [ -n "$K" ]; then CTRLSIM_LIB=$V timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt; fi
if [ -n "$F" ]; then
  echo "== shipped"; timeout 600 python tools/microbench/sustained.py 256 1.5 $F 2>&1 | grep -v amdgpu.ids | tee $O/sustained_shipped.txt
  echo "== variant $2"; CTRLSIM_LIB=$V timeout 600 python tools/microbench/sustained.py 256 1.5 $F 2>&1 | grep -v amdgpu.ids | tee $O/sustained_variant.txt
fi
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile"
for rep in 1 2; do
  timeout 600 $B > $O/a_$rep.json 2> $O/a_$rep.err
  CTRLSIM_LIB=$V timeout 600 $B > $O/b_$rep.json 2> $O/b_$rep.err
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
