#!/bin/bash
# Round 6, final validation of the committed tree (direct-store attention epilogue, model switches, wide contexts in): profile set, GPU suite, smoke(), default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_final2; mkdir -p $O
cd $R
bash tools/profile_round.sh r06 > $O/profile_round.txt 2>&1; tail -2 $O/profile_round.txt | cut -c1-300
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
T0=$SECONDS
timeout 1200 python bench.py --detail-file $O/bench_detail.json > $O/bench_default.json 2> $O/bench_default.err
echo "wall $((SECONDS - T0)) s" | tee $O/wall.txt
wc -c $O/bench_default.json; tail -1 $O/bench_default.json
