#!/bin/bash
# Round 4, VERDICT item 2 (ii): which packed instruction is involved in the co-residency hazard?  Form-1 provocation (simulator step delayed
# 1 ms underneath the other lane's matrix kernels, no guard, CUs shared, 8 scenes) on four builds of sim.hip that differ ONLY in their device
# assembly (tools/probes/build_sim_asm_variant.py): SLP code as clang emits it / v_pk_mov_b32 rewritten / swizzled packed arithmetic rewritten /
# all packed arithmetic rewritten (v_pk_mov_b32 kept).
RUNS=${1:-48}
O=gpurun_out/r04_hazard; mkdir -p $O
export CTRLSIM_SIM_SHARED_CU=1 STRESS_SCENARIOS=8
for v in ${VARIANTS:-slp nopkmov noswz noarith}; do
  t0=$SECONDS
  CTRLSIM_LIB=$PWD/tools/microbench/variants/simv_$v.so timeout 1500 python tools/stress_streams.py $RUNS 0 0 0 111 0 1000 0 1 > $O/stress_${v}_$RUNS.txt 2>&1
  echo "$v: $(tail -1 $O/stress_${v}_$RUNS.txt)  [$((SECONDS - t0)) s]" | tee -a $O/summary.txt
done
if [ -z "$NOPROBE" ]; then (cd tools/probes && timeout 300 ./pk_mfma_probe 3000) > $O/pk_mfma_probe.txt 2>&1; cat $O/pk_mfma_probe.txt; fi
