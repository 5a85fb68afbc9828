#!/bin/bash
# Round 5, job F: attention at 8 waves per workgroup and SIX waves per SIMD (80 VGPRs) against the shipped 4 x 4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_f; mkdir -p $O
cd $R
for v in base _DATT_NW_8_DATT_NW_OCC_6; do
  echo "== $v" | tee -a $O/attn_variants.txt
  CTRLSIM_LIB=$R/tools/microbench/variants/$v.so SUSTAINED_CLASSES=5,8,12,16,20 timeout 600 python tools/microbench/sustained.py 256 1.2 attn+compact 2>&1 | grep -E "attn" | tee -a $O/attn_variants.txt
done
CTRLSIM_LIB=$R/tools/microbench/variants/_DATT_NW_8_DATT_NW_OCC_6.so timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "attention or forward or compact" > $O/pytest_nw8.txt 2>&1; tail -3 $O/pytest_nw8.txt
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile"
for rep in 1 2; do
  timeout 600 $B > $O/a_$rep.json 2> $O/a_$rep.err
  CTRLSIM_LIB=$R/tools/microbench/variants/_DATT_NW_8_DATT_NW_OCC_6.so timeout 600 $B > $O/b_$rep.json 2> $O/b_$rep.err
done
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, round(d["value"]), "spot", d["parity_spot_check"]["identical"], "e2e", round(d["roofline"]["end_to_end"]["frac"],4),
              [(r["kind"], round(r["avg_launch_ms"],4), round(r["frac"],3)) for r in d["roofline"]["kernels"][:7]])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
