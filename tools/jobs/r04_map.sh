#!/bin/bash
O=gpurun_out/r04_map; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "map_pool or seg_emb or test_gpu_model" 2>&1 | tail -3
for rep in 1 2; do
  echo "== packed (shipped)"; timeout 300 python tools/microbench/map_pool.py ${1:-1024} 2>&1 | grep -v amdgpu.ids | tee -a $O/packed.txt
  echo "== scalar FMAs (previous)"; CTRLSIM_LIB=$PWD/tools/microbench/variants/all_mapscalar.so timeout 300 python tools/microbench/map_pool.py ${1:-1024} 2>&1 | grep -v amdgpu.ids | tee -a $O/scalar.txt
done
