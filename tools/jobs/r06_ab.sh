#!/bin/bash
# Round 6, job AB: per-class cost of the causal kernel — mask-table kernel (8 waves, 256-query blocks) against the in-kernel-mask kernel (4 waves, 128-query blocks; option 7 = 0)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ab; mkdir -p $O
cd $R
for v in tbl nw4; do
  if [ $v = tbl ]; then export CTRLSIM_OPTIONS=""; else export CTRLSIM_OPTIONS="7=0"; fi
  timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 0 --fallback-slice 0 --detail-file $O/d_$v.json > $O/b_$v.json 2> $O/err_$v.txt
  python - $O/d_$v.json $v <<'PY' | tee -a $O/classes.txt
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[2], round(d["value"]), "causal frac", [k["frac"] for k in r["kernels"] if k["kind"]=="attention_causal"])
for c in r["causal_attention_by_size_class"] or []:
    print("  ", c["context_slots"], round(c["cycles_per_1e6_visible_pairs"]/1e6,3), round(c["share_of_workgroup_cycles"],4), c["workgroups"])
PY
done
