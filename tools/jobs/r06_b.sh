#!/bin/bash
# Round 6, job B: instruction-mix probe of the attention inner loop, then the whole GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_b; mkdir -p $O
cd $R
timeout 300 tools/probes/attn_mix_probe > $O/attn_mix_probe.txt 2>&1; cat $O/attn_mix_probe.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
