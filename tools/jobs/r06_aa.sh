#!/bin/bash
# Round 6, job AA: cfg.model.no_actions / use_map / encode_initial_state on the HIP path against the reference fixture; wide closed loop
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_aa; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_sim_ctx.py -q -m gpu -k "model_flags or wide_context" 2>&1 | tail -40 | tee $O/flags.txt
