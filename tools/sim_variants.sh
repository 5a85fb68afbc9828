#!/bin/bash
# Variant builds of csrc/sim.hip for the co-residency hazard experiments: each variant is a complete
# libctrlsim_hip.so under tools/microbench/variants/ (git-ignored; travels to the GPU box), selected with CTRLSIM_LIB=<path>.
#   base   the shipped flags          wcnt0  every s_waitcnt forced to zero (-mllvm -amdgpu-waitcnt-forcezero=1)
#   O1     -O1 instead of -O3          fence  agent-scope release/acquire (L2 write-back + L1 invalidate) at every workgroup barrier
cd "$(dirname "$0")/.." || exit 1
V=tools/microbench/variants; mkdir -p $V
objs=$(ls ctrl-sim_amd/csrc/build/*.o | grep -v "/sim.o")
build() {  # name, opt, extra flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $2 -std=c++17 -fPIC -ffp-contract=off -c ctrl-sim_amd/csrc/sim.hip -o $V/sim_$1.o $3 &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/simv_$1.so $objs $V/sim_$1.o && echo built simv_$1
}
build base -O3 "" &
build wcnt0 -O3 "-mllvm -amdgpu-waitcnt-forcezero=1" &
build O1 -O1 "" &
build fence -O3 "-DSIM_FENCE" &
wait
