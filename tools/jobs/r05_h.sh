#!/bin/bash
# Round 5, job H: map_pool with the workgroup-parallel softmax (time, tests), then the round's final lines: driver command on configs[2], configs[1], tilt sweep, RCCL world 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_h; mkdir -p $O
cd $R
timeout 120 python tools/microbench/map_pool.py 1024 2>&1 | tail -2 | tee $O/map_pool_time.txt
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; tail -c 300 $O/bench_driver_style.json; echo
timeout 600 python bench.py --scenarios 256 --agents 32 --polylines 200 --no-cpu-baseline > $O/bench_configs1.json 2> $O/bench_configs1.err
timeout 900 python bench.py --tilt-sweep --no-cpu-baseline > $O/bench_configs4_1gpu.json 2> $O/bench_configs4_1gpu.err
CTRLSIM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --scenarios 204 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
python - $O <<'PY'
import json,sys
for n in ("driver_style","configs1","configs4_1gpu","rccl_world1"):
    try:
        d=json.loads([l for l in open(f"{sys.argv[1]}/bench_{n}.json") if l.startswith("{")][0])
        print(n, round(d["value"]), d["config"]["workload"][:90], "spot", (d["parity_spot_check"] or {}).get("identical"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "map", d["roofline"]["satellite"]["map_pool"]["avg_launch_ms"])
    except Exception as e:
        print(n, "ERR", e)
PY
