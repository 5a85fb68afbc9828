#!/bin/bash
# Kernel-time picture of the K/V-cached phase (32 steps, 102 scenarios, 2 lanes): rocprofv3 kernel stats of tools/microbench/cached_only.py
O=$PWD/gpurun_out/r04_cached; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o cached --output-format csv -- python tools/microbench/cached_only.py 102 2 3 > $O/run.log 2>&1)
cd $GRAFT_REPO_ROOT
grep "cached phase" $O/run.log
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - $F <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} dispatches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.1f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
