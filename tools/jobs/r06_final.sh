#!/bin/bash
# Round 6, final validation of the committed tree: GPU suite, smoke(), the driver's default bench line (stdout kept as the driver would see it)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_final; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
/usr/bin/time -v timeout 1200 python bench.py --detail-file $O/bench_detail.json > $O/bench_default.json 2> $O/bench_default.err
grep -E "Elapsed \(wall" $O/bench_default.err | tee $O/wall.txt
wc -c $O/bench_default.json; tail -1 $O/bench_default.json
