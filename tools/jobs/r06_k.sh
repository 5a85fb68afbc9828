#!/bin/bash
# Round 6, job K: the other BASELINE shapes on one GPU with the final build (configs[1], the configs[4] tilt sweep, RCCL at world size 1),
# the plugin routes, and `python bench.py --gpus 2` WITHOUT a launcher on the one GPU (self-launch; both ranks share the device, gloo collectives)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_k; mkdir -p $O
cd $R
timeout 600 python bench.py --scenarios 256 --agents 32 --polylines 200 --no-cpu-baseline --fallback-slice 0 --detail-file $O/bench_configs1.json > $O/short_configs1.json 2> $O/bench_configs1.err; cat $O/short_configs1.json | cut -c1-400
timeout 900 python bench.py --tilt-sweep --no-cpu-baseline --fallback-slice 0 --detail-file $O/bench_configs4_1gpu.json > $O/short_configs4.json 2> $O/bench_configs4_1gpu.err; cat $O/short_configs4.json | cut -c1-400
CTRLSIM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --scenarios 204 --steps 2 --warmup 1 --no-cpu-baseline --fallback-slice 0 --detail-file $O/bench_rccl_world1.json > $O/short_rccl.json 2> $O/bench_rccl_world1.err; cat $O/short_rccl.json | cut -c1-300
CTRLSIM_BENCH_DEBUG_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --scenarios 24 --steps 2 --warmup 0 --max-ctx 128 --no-cpu-baseline --fallback-slice 0 --detail-file $O/bench_selflaunch2.json > $O/short_selflaunch2.json 2> $O/bench_selflaunch2.err; echo "self-launch rc $?"; cat $O/short_selflaunch2.json | cut -c1-300
timeout 600 python tools/facade_rate.py 7 8 90 2>&1 | tail -1 | tee $O/facade_rate.txt
timeout 600 python tools/facade_rate.py 63 8 90 batched 2>&1 | tail -2 | tee -a $O/facade_rate.txt
timeout 600 python tools/facade_rate.py 255 8 90 batched 2>&1 | tail -2 | tee -a $O/facade_rate.txt
