#!/bin/bash
# Round 5, job Z: rate of the row-per-lane 16-byte access pattern (32 lines per wave instruction) against the coalesced one, from L2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_z; mkdir -p $O
cd $R/tools/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/row_load_probe row_load_probe.hip 2>/dev/null
/tmp/row_load_probe | tee $O/row_load_probe.txt
