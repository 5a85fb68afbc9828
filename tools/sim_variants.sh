#!/bin/bash
# Variant builds of csrc/sim.hip for the co-residency hazard experiments: each variant is a complete
# libctrlsim_hip.so under tools/microbench/variants/ (git-ignored; travels to the GPU box), selected with CTRLSIM_LIB=<path>.
# usage: tools/sim_variants.sh [name "<opt level>" "<extra flags>"]...   (no arguments: the round-3 set)
cd "$(dirname "$0")/.." || exit 1
V=tools/microbench/variants; mkdir -p $V
objs=$(ls ctrl-sim_amd/csrc/build/*.o | grep -v "/sim.o")
build() {  # name, opt, extra flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $2 -std=c++17 -fPIC -ffp-contract=off -c ctrl-sim_amd/csrc/sim.hip -o $V/sim_$1.o $3 &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/simv_$1.so $objs $V/sim_$1.o && echo built simv_$1
}
if [ $# -gt 0 ]; then
  while [ $# -gt 0 ]; do build "$1" "$2" "$3" & shift 3; done
else
  build base -O3 "" &
  build O1 -O1 "" &
  build O2 -O2 "" &
  build noalias -O3 "-fno-strict-aliasing" &
  build novec -O3 "-fno-slp-vectorize -fno-vectorize -mllvm -amdgpu-load-store-vectorizer=0" &
  build nosched -O3 "-mllvm -enable-misched=0 -mllvm -enable-post-misched=0" &
fi
wait
