#!/bin/bash
# Round-4 artifacts, part B: the driver's command on configs[2], the other configurations, the plugin-route rate.
O=gpurun_out; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_driver_style.json 2> $O/r04_bench_driver_style.err; tail -c 300 $O/r04_bench_driver_style.json; echo
timeout 600 python bench.py --scenarios 256 --agents 32 --polylines 200 --no-cpu-baseline > $O/r04_bench_configs1.json 2> $O/r04_bench_configs1.err
timeout 900 python bench.py --tilt-sweep --no-cpu-baseline > $O/r04_bench_configs4_1gpu.json 2> $O/r04_bench_configs4_1gpu.err
CTRLSIM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --scenarios 204 --steps 2 --warmup 1 --no-cpu-baseline > $O/r04_bench_rccl_world1.json 2> $O/r04_bench_rccl_world1.err
timeout 300 python tools/facade_rate.py > $O/r04_facade_rate.txt 2>&1
python - <<'PY'
import json
for n in ("driver_style","configs1","configs4_1gpu","rccl_world1"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r04_bench_{n}.json") if l.startswith("{")][0])
        print(n, round(d["value"]), d["config"]["workload"][:90], "spot", (d["parity_spot_check"] or {}).get("identical"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "ERR", e)
PY
tail -2 $O/r04_facade_rate.txt
bash tools/jobs/r04_hazard_final.sh 96 200
