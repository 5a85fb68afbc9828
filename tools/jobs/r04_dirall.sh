#!/bin/bash
# Every attention launch in the streaming form (option 9 = 2; one wave per 32 queries, K/V fragments from L2): tests, sustained rows, rollout A/B.
O=gpurun_out/r04_dirall; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_default.txt 2>&1; tail -2 $O/pytest_default.txt
CTRLSIM_OPTIONS=9=2 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "attention or forward or model" > $O/pytest_dirall.txt 2>&1; tail -2 $O/pytest_dirall.txt
echo "== staged"; SUSTAINED_CLASSES=8,16 timeout 600 python tools/microbench/sustained.py 256 1.0 attn+compact 2>&1 | grep -v amdgpu.ids | tee $O/sustained_a.txt
echo "== streaming everywhere"; CTRLSIM_OPTIONS=9=2 SUSTAINED_CLASSES=8,16 timeout 600 python tools/microbench/sustained.py 256 1.0 attn+compact 2>&1 | grep -v amdgpu.ids | tee $O/sustained_b.txt
B="python bench.py --scenarios 408 --steps 4 --warmup 1 --no-cpu-baseline --spot-check 2 --no-class-profile"
timeout 600 $B > $O/a_1.json 2> $O/a_1.err
CTRLSIM_OPTIONS=9=2 timeout 600 $B > $O/b_1.json 2> $O/b_1.err
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, round(d["value"]), "spot", d["parity_spot_check"]["identical"], "e2e", round(d["roofline"]["end_to_end"]["frac"],4),
              [(r["kernel"][:14], round(r["avg_launch_ms"],4), round(r["frac"],3)) for r in d["roofline"]["kernels"][:6]])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
