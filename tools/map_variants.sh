#!/bin/bash
# ablation builds of csrc/map_encoder.hip (complete libraries under tools/microbench/variants/, CTRLSIM_LIB selects one)
cd "$(dirname "$0")/.." || exit 1
V=tools/microbench/variants; mkdir -p $V
objs=$(ls ctrl-sim_amd/csrc/build/*.o | grep -v "/map_encoder.o")
build() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c ctrl-sim_amd/csrc/map_encoder.hip -o $V/map_$1.o $2 &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/mapv_$1.so $objs $V/map_$1.o && echo built mapv_$1
}
build base "" &
build nop3 "-DMP_ABL_NO_P3" &
build nop12 "-DMP_ABL_NO_P1 -DMP_ABL_NO_P2" &
build nop123 "-DMP_ABL_NO_P1 -DMP_ABL_NO_P2 -DMP_ABL_NO_P3" &
wait
