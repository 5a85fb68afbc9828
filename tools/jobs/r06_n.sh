#!/bin/bash
# Round 6, job N: the round's profile set re-taken on the FINAL build (resident-keys cross attention in)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_n; mkdir -p $O
cd $R
bash tools/profile_round.sh r06 > $O/profile_round.txt 2>&1; tail -3 $O/profile_round.txt | cut -c1-400
