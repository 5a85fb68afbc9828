#!/bin/bash
# Round 5, job N: where the QP kernel's time goes — store placement variants (tools/microbench/variants/all_qp*.so), same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_n; mkdir -p $O
cd $R
for v in B A C D E F G; do
  if [ $v = B ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_qp$v.so; fi
  echo "$v: $(CTRLSIM_LIB=$L timeout 120 python tools/microbench/qp_timing.py 2>&1 | tail -1)" | tee -a $O/qp_timing.txt
done
for v in A B A B; do
  if [ $v = B ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/all_qp$v.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile > $O/bench_$v.$RANDOM.json 2> $O/bench_err.txt
done
CTRLSIM_OPTIONS=3=2 timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile > $O/bench_opt2.json 2> $O/bench_err.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"])
PY
done
