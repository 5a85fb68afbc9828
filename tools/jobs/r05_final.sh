#!/bin/bash
# Round 5, final validation of the committed tree: GPU suite, smoke(), the driver's default bench line, hazard provocations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_final; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
( export CTRLSIM_SIM_SHARED_CU=1 STRESS_SCENARIOS=8
  timeout 900 python tools/stress_streams.py 64 0 0 0 111 0 1000 0 1 > $O/hazard_form1.txt 2>&1; echo "form 1: $(tail -1 $O/hazard_form1.txt)" | tee -a $O/hazard_summary.txt
  timeout 1200 python tools/stress_streams.py 96 1 1 1 111 0 0 0 1 > $O/hazard_form2.txt 2>&1; echo "form 2: $(tail -1 $O/hazard_form2.txt)" | tee -a $O/hazard_summary.txt )
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print("default bench:", round(d["value"]), d["steps"], d["warmup"], round(d["ms_per_step"]), "roofline", round(d["roofline"]["frac"],3), d["roofline"]["traffic_source"], "cpu", d["cpu_baseline"]["value"], "spot", d["parity_spot_check"]["identical"])
PY
