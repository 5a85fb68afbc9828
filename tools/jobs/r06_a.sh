#!/bin/bash
# Round 6, job A: baseline of the round — the whole GPU suite, then the driver's command (short stdout line + detail file)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_a; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc $?"
wc -c $O/bench_stdout.txt; cat $O/bench_stdout.txt
cp bench_detail.json $O/bench_detail.json 2>/dev/null
