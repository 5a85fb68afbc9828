#!/bin/bash
# Round-4 artifacts, part A: the whole GPU test suite, the profiled bench command (rocprofv3 kernel stats + serialised twin + PMC passes).
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04_pytest_gpu.txt 2>&1; tail -3 $O/r04_pytest_gpu.txt
timeout 1500 bash tools/profile_round.sh r04 > $O/r04_profile_round.log 2>&1; tail -5 $O/r04_profile_round.log
