#!/bin/bash
# Round 6, job V: profile set re-taken on the build with the packed-fp32 map_pool kernel and the own-return mask mode; SQ counters of map_pool_pk_kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_v; mkdir -p $O
cd $R
bash tools/profile_round.sh r06 > $O/profile_round.txt 2>&1; tail -3 $O/profile_round.txt | cut -c1-400
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $set -d $O/map$i -o pmc --output-format csv -- python tools/microbench/pmc_map_pool.py > $O/map$i.log 2>&1)
  f=$(find $O/map$i -name "*counter_collection.csv" | head -1)
  echo "== map_pool set $i"
  python - "$f" <<'PY' | tee -a $O/map_pool_counters.txt
import csv,sys,collections
agg=collections.defaultdict(float); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "map_pool" not in r['Kernel_Name']: continue
    agg[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
print({c: f"{v/cnt[c]:.6g}" for c,v in sorted(agg.items())}, "dispatches", max(cnt.values()) if cnt else 0)
PY
done
find $O -name "*counter_collection.csv" -size +2M -delete
(cd $R && timeout 120 python tools/microbench/map_pool.py 1024 2>&1 | tail -3 | tee $O/map_pool_time.txt)
