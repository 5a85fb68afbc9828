#!/bin/bash
# Round 4: causal attention with per-class visibility-mask tables (option 7) — correctness first, then same-box A/B numbers.
O=gpurun_out/r04_attn; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "mask_tables or structured_causal or presplit_images" > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
if [ -z "$SKIP_ALL" ]; then timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_all.txt 2>&1; tail -5 $O/pytest_all.txt; fi
export SUSTAINED_CLASSES=${SUSTAINED_CLASSES:-4,8,12,16,20}
timeout 600 python tools/microbench/sustained.py 256 1.5 attn+compact > $O/sustained_default.txt 2>&1; cat $O/sustained_default.txt
for v in ${VARIANTS-noearly}; do
  CTRLSIM_LIB=$PWD/tools/microbench/variants/all_$v.so timeout 600 python tools/microbench/sustained.py 256 1.5 attn+compact > $O/sustained_$v.txt 2>&1
  echo "== variant $v"; cat $O/sustained_$v.txt
done
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2"
for rep in 1 2; do
  CTRLSIM_OPTIONS=7=0 timeout 600 $B --no-pipeline > $O/bench_tbl0_$rep.json 2> $O/bench_tbl0_$rep.err
  timeout 600 $B --no-pipeline > $O/bench_tbl1_$rep.json 2> $O/bench_tbl1_$rep.err
  timeout 600 $B > $O/bench_tbl1p_$rep.json 2> $O/bench_tbl1p_$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_attn/bench_tbl*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        k=[r for r in d["roofline"]["kernels"] if r["kernel"].startswith("causal")]
        print(f, round(d["value"]), d["parity_spot_check"]["identical"], d["config"].get("phases"), "attn class frac", round(d["roofline"]["other"]["frac"],4), "causal main", [(round(r["frac"],4), round(r["avg_launch_ms"],4)) for r in k])
    except Exception as e:
        print(f, "ERR", e)
PY
