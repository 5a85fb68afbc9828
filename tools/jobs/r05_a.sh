#!/bin/bash
# Round 5, job A: baselines on one box + the counter evidence the round-4 review asked for.
#  1. sustained kernel rates (FFN / Linear / attention) of the round-4 build
#  2. FETCH_SIZE / WRITE_SIZE calibration: access-pattern probes with known bytes + the four hot kernels at known-byte launches
#  3. SQ counters of map_pool_kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_a; mkdir -p $O
cd $R
timeout 600 python tools/microbench/sustained.py 256 1.5 > $O/sustained.txt 2>&1; cat $O/sustained.txt | tail -20
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout 300 rocprofv3 --pmc $c -d $O/probe_$c -o pmc --output-format csv -- tools/probes/pmc_calib 1024 3 > $O/probe_$c.log 2>&1)
  (cd $R && timeout 600 rocprofv3 --pmc $c -d $O/kern_$c -o pmc --output-format csv -- python tools/microbench/pmc_calib_kernels.py > $O/kern_$c.log 2>&1)
done
grep -h '^{' $O/probe_FETCH_SIZE.log | tail -1 > $O/probe_expect.json
grep -h '^{' $O/kern_FETCH_SIZE.log | tail -1 > $O/kern_expect.json
(cd $R && python tools/pmc_calib_summary.py $O > $O/calib_summary.json 2> $O/calib_summary.err; tail -5 $O/calib_summary.err; head -c 6000 $O/calib_summary.json)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $set -d $O/map$i -o pmc --output-format csv -- python tools/microbench/pmc_map_pool.py > $O/map$i.log 2>&1)
  f=$(find $O/map$i -name "*counter_collection.csv" | head -1)
  echo "== map_pool set $i"
  python - "$f" <<'PY' | tee -a $O/map_pool_counters.txt
import csv,sys,collections
agg=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "map_pool_kernel" not in r['Kernel_Name']: continue
    agg[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
print({c: f"{v/cnt[c]:.6g}" for c,v in sorted(agg.items())}, "dispatches", max(cnt.values()) if cnt else 0)
PY
done
find $O -name "*counter_collection.csv" -size +2M -delete
(cd $R && timeout 120 python tools/microbench/map_pool.py 1024 2>&1 | tail -3 | tee $O/map_pool_time.txt)
