#!/bin/bash
# builds variants of one split-operand source into tools/microbench/variants/<name>.so (other objects from csrc/build)
# usage: mkvariants.sh <source without .hip> "<defs variant 1>" "<defs variant 2>" ...   ("" = base)
SRC=$1; shift
mkdir -p tools/microbench/variants
cd /root/repo
for d in "$@"; do
  name=$(echo "$d" | tr -d ' ' | tr -c 'A-Za-z0-9_\n' '_'); [ -z "$name" ] && name=base
  ( ok=1
    for sch in 1 0; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c ctrl-sim_amd/csrc/$SRC.hip -o tools/microbench/variants/${name}_s$sch.o -DCTRLSIM_F16X3=$sch $d 2>/dev/null || ok=0
    done
    objs=$(ls ctrl-sim_amd/csrc/build/*.o | grep -v "/${SRC}_s[01].o")
    [ $ok = 1 ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/microbench/variants/$name.so $objs tools/microbench/variants/${name}_s1.o tools/microbench/variants/${name}_s0.o && echo built $name ) &
done
wait
