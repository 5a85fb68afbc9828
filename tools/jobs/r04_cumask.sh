#!/bin/bash
# Round 4, VERDICT item 7: few-row kernels (side streams) on a fixed eighth / quarter of the compute units, full-row kernels on the rest.
O=gpurun_out/r04_cumask; mkdir -p $O
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile"
S8=01010101; M8=fefefefe          # striped: one compute unit in eight for the side streams
S4=11111111; M4=eeeeeeee          # striped: one in four
rep8() { printf "%s,%s,%s,%s,%s,%s,%s,%s" $1 $1 $1 $1 $1 $1 $1 $1; }
run() { name=$1; shift; env "$@" timeout 900 $B > $O/$name.json 2> $O/$name.err; python - $O/$name.json <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f, round(d["value"]), "spot", d["parity_spot_check"]["identical"], "e2e", round(d["roofline"]["end_to_end"]["frac"],4), [(r["kernel"][:10], round(r["avg_launch_ms"],3)) for r in d["roofline"]["kernels"][:4]])
except Exception as e:
    print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
}
run base X=1
run stripe8 CTRLSIM_SIDE_CU_MASK=$(rep8 $S8) CTRLSIM_MAIN_CU_MASK=$(rep8 $M8)
run stripe8_sideonly CTRLSIM_SIDE_CU_MASK=$(rep8 $S8)
run stripe4 CTRLSIM_SIDE_CU_MASK=$(rep8 $S4) CTRLSIM_MAIN_CU_MASK=$(rep8 $M4)
run block8 CTRLSIM_SIDE_CU_MASK=ffffffff,0,0,0,0,0,0,0 CTRLSIM_MAIN_CU_MASK=0,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff
run base2 X=1
