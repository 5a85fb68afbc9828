#!/bin/bash
# Round 6, job H: image tails (one zeroing launch per pass for the scene-side image sets, whole-sub-tile tails not zeroed) under a poisoned
# workspace; then the round's profile set (bench line, rocprofv3 kernel stats, serialised twin, PMC traffic)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_h; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_sim_ctx.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
bash tools/profile_round.sh r06 > $O/profile_round.txt 2>&1; tail -5 $O/profile_round.txt
