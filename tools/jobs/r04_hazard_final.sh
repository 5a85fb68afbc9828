#!/bin/bash
# Round 4, VERDICT item 2 (iii): provoked runs of both forms of the co-residency hazard on the SHIPPED library (tools/stress_streams.py;
# DESIGN.md section 4).  Form 1: simulator step delayed 1 ms underneath the other lane's matrix kernels.  Form 2: second pass, first-pass
# tail and cached steps on the side streams underneath full-row kernels.  No stream guard, CUs shared, 8 full-size scenes.
R1=${1:-192}; R2=${2:-320}
O=gpurun_out/r04_hazard; mkdir -p $O
export CTRLSIM_SIM_SHARED_CU=1 STRESS_SCENARIOS=8
unset CTRLSIM_LIB
t0=$SECONDS; timeout 1500 python tools/stress_streams.py $R1 0 0 0 111 0 1000 0 1 > $O/final_form1_$R1.txt 2>&1
echo "shipped, form 1: $(tail -1 $O/final_form1_$R1.txt)  [$((SECONDS - t0)) s]" | tee -a $O/summary.txt
t0=$SECONDS; timeout 2400 python tools/stress_streams.py $R2 1 1 1 111 0 0 0 1 > $O/final_form2_$R2.txt 2>&1
echo "shipped, form 2: $(tail -1 $O/final_form2_$R2.txt)  [$((SECONDS - t0)) s]" | tee -a $O/summary.txt
