#!/bin/bash
# Round 6, job AC: per class, alone on the device: in-kernel-mask causal kernel (4 waves, 128-query blocks) against the mask-table kernel (8 waves, 256-query blocks)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ac; mkdir -p $O
cd $R
SUSTAINED_CLASSES=4,5,6,7,8,9,10,11,12,14,16,20 timeout 900 python tools/microbench/sustained.py 256 0.8 "compact" 2>&1 | grep -E "^attn" | tee $O/classes.txt
