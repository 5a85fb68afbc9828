#!/bin/bash
# Round 6, job O: validation after the batched fill_index launch — whole GPU suite + a rollout line with the spot check
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_o; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 3 --no-class-profile --fallback-slice 0 --detail-file $O/d.json > $O/b.json 2> $O/bench_err.txt; cut -c1-420 $O/b.json; grep -o '"parity_spot_check": {[^}]*}' $O/b.json
