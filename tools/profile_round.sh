#!/bin/bash
# One GPU-box call that produces everything tools/make_profile_summary.py needs for round $1 (e.g. r02):
#   an un-profiled bench line, the rocprofv3 --kernel-trace --stats summary of the same command, and the two PMC passes
#   (FETCH_SIZE / WRITE_SIZE, separate, no tracing flags) over a smaller run of the same workload.
# usage (from the repo root, on the GPU box):  bash tools/profile_round.sh r02
R=${1:-r02}
CMD="python bench.py --scenarios 204 --steps 2 --warmup 1 --spot-check 0 --no-class-profile --fallback-slice 0"
PMC="python bench.py --scenarios 102 --steps 1 --warmup 0 --no-cpu-baseline --spot-check 0 --no-class-profile --fallback-slice 0 --detail-file gpurun_out/pmc_$R/detail.json"
mkdir -p gpurun_out/prof_$R gpurun_out/pmc_$R
$CMD --detail-file gpurun_out/bench_$R.json > gpurun_out/bench_$R.short.json 2> gpurun_out/bench_$R.err     # (the detail file is what the summaries read)
echo "$CMD --no-cpu-baseline" > gpurun_out/prof_$R/command.txt
echo "$PMC" > gpurun_out/pmc_$R/command.txt
ROOT=$(pwd)
( cd /tmp && export TMPDIR=/tmp && cd "$ROOT" && rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R -o $R --output-format csv -- $CMD --no-cpu-baseline --fallback-slice 0 --detail-file gpurun_out/prof_$R/detail.json > gpurun_out/prof_$R/run.log 2>&1 )
find gpurun_out/prof_$R -name "*_kernel_trace.csv" -delete          # tens of MB; the stats csv is what is summarised
find gpurun_out/prof_$R -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_$R/${R}_kernel_stats.csv \; 2>/dev/null
# the same command with every kernel on ONE stream (--side ""): kernel durations then do not overlap and rocprofv3's per-kernel averages
# can be compared one to one with bench.py's HIP-event averages (with the side streams on, an event interval on a side stream also
# holds the kernel's wait for free CUs)
mkdir -p gpurun_out/prof_${R}s
$CMD --side "" --no-cpu-baseline --detail-file gpurun_out/bench_${R}s.json > gpurun_out/bench_${R}s.short.json 2> gpurun_out/bench_${R}s.err
( cd /tmp && export TMPDIR=/tmp && cd "$ROOT" && rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${R}s -o ${R}s --output-format csv -- $CMD --side "" --no-cpu-baseline --fallback-slice 0 --detail-file gpurun_out/prof_${R}s/detail.json > gpurun_out/prof_${R}s/run.log 2>&1 )
find gpurun_out/prof_${R}s -name "*_kernel_trace.csv" -delete
find gpurun_out/prof_${R}s -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_${R}s/${R}s_kernel_stats.csv \; 2>/dev/null
bash tools/pmc_traffic.sh gpurun_out/pmc_$R -- $PMC > /dev/null
find gpurun_out/pmc_$R -name "*counter_collection.csv" -delete      # summarised in summary.json
ls -la gpurun_out/prof_$R gpurun_out/pmc_$R; cat gpurun_out/bench_$R.short.json
