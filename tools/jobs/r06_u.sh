#!/bin/bash
# Round 6, job U: the build with the packed-fp32 map_pool kernel — map tests, full GPU suite (hazard tests included), co-residency provocations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_u; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "map" 2>&1 | tail -3 | tee $O/ops.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
( export CTRLSIM_SIM_SHARED_CU=1 STRESS_SCENARIOS=8
  timeout 900 python tools/stress_streams.py 40 0 0 0 111 0 1000 0 1 > $O/hazard_form1.txt 2>&1; echo "form 1: $(tail -1 $O/hazard_form1.txt)" | tee -a $O/hazard_summary.txt
  timeout 1200 python tools/stress_streams.py 56 1 1 1 111 0 0 0 1 > $O/hazard_form2.txt 2>&1; echo "form 2: $(tail -1 $O/hazard_form2.txt)" | tee -a $O/hazard_summary.txt )
