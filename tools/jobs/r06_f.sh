#!/bin/bash
# Round 6, job F: is the cost of the K/V image DMA (16-22 % of the attention kernels although its latency is hidden) a CLOCK effect?
# shader clock + socket power sampled while the causal kernel runs back to back, with and without its DMA
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_f; mkdir -p $O
cd $R
for v in new _DATT_ABL_NODMA; do
  if [ $v = new ]; then L=$R/ctrl-sim_amd/csrc/libctrlsim_hip.so; else L=$R/tools/microbench/variants/$v.so; fi
  echo "== $v" | tee -a $O/clocks.txt
  ( CTRLSIM_LIB=$L timeout 120 python tools/microbench/sustained.py 256 4.0 "attn" 2>&1 | grep -E "^attn (causal mask|cross)" > $O/s_$v.txt ) &
  PID=$!
  sleep 14
  for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $O/clocks.txt; echo >> $O/clocks.txt
    sleep 1
  done
  wait $PID
  cat $O/s_$v.txt | tee -a $O/clocks.txt
done
cat $O/clocks.txt | cut -c1-230
