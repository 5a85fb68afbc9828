#!/bin/bash
# Round 5, job S: final lines with both out-projection fusions on: GPU suite, hazard provocations on the shipped build, driver command,
# configs[1], tilt sweep, RCCL world 1, kernel trace, per-kernel PMC traffic
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_s; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
( export CTRLSIM_SIM_SHARED_CU=1 STRESS_SCENARIOS=8; unset CTRLSIM_LIB
  t0=$SECONDS; timeout 900 python tools/stress_streams.py 64 0 0 0 111 0 1000 0 1 > $O/hazard_form1.txt 2>&1; echo "form 1: $(tail -1 $O/hazard_form1.txt) [$((SECONDS - t0)) s]" | tee -a $O/hazard_summary.txt
  t0=$SECONDS; timeout 1200 python tools/stress_streams.py 96 1 1 1 111 0 0 0 1 > $O/hazard_form2.txt 2>&1; echo "form 2: $(tail -1 $O/hazard_form2.txt) [$((SECONDS - t0)) s]" | tee -a $O/hazard_summary.txt )
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; tail -c 300 $O/bench_driver_style.json; echo
timeout 600 python bench.py --scenarios 256 --agents 32 --polylines 200 --no-cpu-baseline > $O/bench_configs1.json 2> $O/bench_configs1.err
timeout 900 python bench.py --tilt-sweep --no-cpu-baseline > $O/bench_configs4_1gpu.json 2> $O/bench_configs4_1gpu.err
CTRLSIM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --scenarios 204 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err
python - $O <<'PY'
import json,sys
for n in ("driver_style","configs1","configs4_1gpu","rccl_world1"):
    try:
        d=json.loads([l for l in open(f"{sys.argv[1]}/bench_{n}.json") if l.startswith("{")][0])
        print(n, round(d["value"]), d["config"]["workload"][:90], "spot", (d["parity_spot_check"] or {}).get("identical"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "linear frac", d["roofline"]["frac"])
    except Exception as e:
        print(n, "ERR", e)
PY
( cd /tmp && export TMPDIR=/tmp && cd $R && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o r05u --output-format csv -- python bench.py --scenarios 204 --steps 2 --warmup 1 --spot-check 0 --no-class-profile --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err )
find $O/prof -name "*_kernel_trace.csv" -delete
timeout 900 python bench.py --scenarios 204 --steps 2 --warmup 1 --spot-check 0 --no-class-profile > $O/unprofiled_bench.json 2> $O/unprofiled.err
timeout 1500 bash tools/pmc_traffic.sh $O/pmc -- python bench.py --scenarios 204 --steps 2 --warmup 1 --spot-check 0 --no-class-profile --no-cpu-baseline 2>&1 | tail -12
find $O/pmc -name "*counter_collection.csv" -size +20M -delete
