#!/bin/bash
# Round 5, job B: MFMA chain probe, fused-FFN variants (two hacc chains / kk-major second product), full GPU test suite after the guard-pair refactor
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_b; mkdir -p $O
cd $R
timeout 300 tools/probes/mfma_chain_probe 4000 > $O/chain_probe.txt 2>&1; cat $O/chain_probe.txt
for v in base _DFFN_H2 _DFFN_KKMAJOR _DFFN_H2_DFFN_KKMAJOR; do
  echo "== $v" | tee -a $O/ffn_variants.txt
  CTRLSIM_LIB=$R/tools/microbench/variants/$v.so timeout 300 python tools/microbench/sustained.py 256 1.5 ffn 2>&1 | grep "ffn" | tee -a $O/ffn_variants.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
