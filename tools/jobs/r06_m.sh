#!/bin/bash
# Round 6, job M: validation of the committed tree — whole GPU suite, smoke(), the driver's command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_m; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $O/bench_detail_configs2.json > $O/bench_configs2.json 2> $O/bench_configs2.err; echo "bench rc $?"; cat $O/bench_configs2.json
( cd /tmp && export TMPDIR=/tmp && cd $R && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o m --output-format csv -- python bench.py --scenarios 204 --steps 2 --warmup 1 --spot-check 0 --no-class-profile --fallback-slice 0 --no-cpu-baseline --detail-file $O/prof_detail.json > $O/prof_run.log 2>&1 )
find $O/prof -name "*_kernel_trace.csv" -delete; find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/m_kernel_stats.csv \;
python - $O/m_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:14]: print(r["Name"][:70], r["Calls"], round(float(r["TotalDurationNs"])/1e6,1), round(100*float(r["TotalDurationNs"])/tot,2))
print("total ms", round(tot/1e6), "dispatches", sum(int(r["Calls"]) for r in rows))
for r in rows:
    if "zero_tails" in r["Name"] or "row_copy" in r["Name"]: print(r["Name"][:60], r["Calls"], round(float(r["TotalDurationNs"])/1e6,1))
PY
