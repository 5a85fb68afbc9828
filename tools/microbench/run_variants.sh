#!/bin/bash
# usage: run_variants.sh <bench script> [args]; runs it once per tools/microbench/variants/*.so
cp ctrl-sim_amd/csrc/libctrlsim_hip.so /tmp/orig.so
for v in tools/microbench/variants/*.so; do
  cp $v ctrl-sim_amd/csrc/libctrlsim_hip.so
  echo "== $(basename $v .so)"; timeout 900 python "$@" 2>&1 | grep -E "^(gemm|attn|impl|ffn|map|\{)"
done
cp /tmp/orig.so ctrl-sim_amd/csrc/libctrlsim_hip.so
