#!/bin/bash
# Round 6, job T: map_pool with packed fp32 FMAs (product library) against the scalar kernel (-DMAP_PK=0 variant): op parity, bits, time, rollout pair
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_t; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "map" 2>&1 | tail -3 | tee $O/ops.txt
for v in pk scalar pk scalar; do
  if [ $v = pk ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/map_scalar.so; fi
  CTRLSIM_LIB=$L timeout 300 python tools/microbench/map_ab.py $O/map_$v.npz 2>&1 | grep map_pool | sed "s/^/$v /" | tee -a $O/ab.txt
done
python - $O <<'PY' | tee -a $O/ab.txt
import numpy as np, sys
a = np.load(sys.argv[1] + "/map_pk.npz"); b = np.load(sys.argv[1] + "/map_scalar.npz")
for t in ("ragged", "full"):
    x, y = a[t + "_out"], b[t + "_out"]
    print(t, "rows", x.shape[0], "bit-identical rows", int((x.view(np.uint32) == y.view(np.uint32)).all(1).sum()), "max abs diff", float(np.abs(x - y).max()),
          "pad equal", bool((a[t + "_pad"] == b[t + "_pad"]).all()), "finite", bool(np.isfinite(x).all()))
PY
for v in pk scalar pk scalar; do
  if [ $v = pk ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/map_scalar.so; fi
  CTRLSIM_LIB=$L timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile --fallback-slice 0 --detail-file $O/d_$v.json > $O/b.json 2> $O/bench_err.txt
  python - $O/b.json $v <<'PY' | tee -a $O/ab.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],1), d.get("sclk_mhz"), d.get("socket_power_w"), d["parity_spot_check"])
PY
done
