#!/bin/bash
# Round 5, job U: the shipped default (option 3 = 2): new tests, driver command, kernel trace, per-kernel PMC traffic
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_u2; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; tail -c 200 $O/bench_driver_style.json; echo
( cd /tmp && export TMPDIR=/tmp && cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o r05w --output-format csv -- python bench.py --scenarios 204 --steps 2 --warmup 1 --spot-check 0 --no-class-profile --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err )
find $O/prof -name "*_kernel_trace.csv" -delete
timeout 900 python bench.py --scenarios 204 --steps 2 --warmup 1 --spot-check 0 --no-class-profile > $O/unprofiled_bench.json 2> $O/unprofiled.err
timeout 1500 bash tools/pmc_traffic.sh $O/pmc -- python bench.py --scenarios 204 --steps 2 --warmup 1 --spot-check 0 --no-class-profile --no-cpu-baseline 2>&1 | tail -8
timeout 600 python bench.py --scenarios 256 --agents 32 --polylines 200 --no-cpu-baseline > $O/bench_configs1.json 2> $O/bench_configs1.err
timeout 900 python bench.py --tilt-sweep --no-cpu-baseline > $O/bench_configs4_1gpu.json 2> $O/bench_configs4_1gpu.err
CTRLSIM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --scenarios 204 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
python - $O <<'PY'
import json,sys
for n in ("driver_style","configs1","configs4_1gpu","rccl_world1"):
    try:
        d=json.loads([l for l in open(f"{sys.argv[1]}/bench_{n}.json") if l.startswith("{")][0])
        print(n, round(d["value"]), d["config"]["workload"][:60], "spot", (d["parity_spot_check"] or {}).get("identical"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "linear frac", round(d["roofline"]["frac"],4))
    except Exception as e:
        print(n, "ERR", e)
PY
