#!/bin/bash
# Round 6, job G: map_pool with its thread-per-point phase on all four waves (compile-time trip counts) against the round-5 kernel; the tree
# (control-word attention, scattered-row Linear, no fp32 K/V scatter) against round 5's attention control flow; whole GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_g; mkdir -p $O
cd $R
for v in new mapold new mapold; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  echo "$v $(CTRLSIM_LIB=$L timeout 300 python tools/microbench/map_pool.py 1024 2>&1 | grep map_pool | tr '\n' ' ')" | tee -a $O/map_pool.txt
done
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "map" > $O/pytest_map.txt 2>&1; tail -2 $O/pytest_map.txt
for v in new r5flow new r5flow; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 0 --no-class-profile --fallback-slice 0 --detail-file $O/d_$v.json > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],1), d["roofline"]["attention_causal_frac"])
PY
done
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
