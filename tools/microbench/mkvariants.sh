#!/bin/bash
# builds ablation variants of one source into tools/microbench/variants/<name>.so (other objects from csrc/build)
SRC=$1; shift
mkdir -p tools/microbench/variants
cd /root/repo
i=0
for d in "$@"; do
  name=$(echo "$d" | tr -d ' ' | tr -c 'A-Za-z0-9_\n' '_'); [ -z "$name" ] && name=base
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c ctrl-sim_amd/csrc/$SRC.hip -o tools/microbench/variants/$name.o $d 2>/dev/null &&
    objs=$(ls ctrl-sim_amd/csrc/build/*.o | grep -v "/$SRC.o") &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/microbench/variants/$name.so $objs tools/microbench/variants/$name.o && echo built $name ) &
done
wait
