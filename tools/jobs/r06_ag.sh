#!/bin/bash
# Round 6, job AG: durations of the GPU suite (it grew from 396 s to 600 s this round)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ag; mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -x -q --durations=40 > $O/pytest.txt 2>&1; tail -50 $O/pytest.txt
