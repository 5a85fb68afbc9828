#!/bin/bash
# Round 6, job AJ: SQ counters of the causal attention kernel on the final build (the round-5 passes of r05_e.sh: instructions per MFMA after the control words)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_aj; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $set -d $O/att$i -o pmc --output-format csv -- python tools/microbench/pmc_attn.py > $O/att$i.log 2>&1)
  f=$(find $O/att$i -name "*counter_collection.csv" | head -1)
  echo "== attention set $i" | tee -a $O/attn_counters.txt
  python - "$f" <<'PY' | tee -a $O/attn_counters.txt
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
try:
    rows=list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no csv", e); rows=[]
for r in rows:
    if "attention_bf16x6_kernel" not in r['Kernel_Name']: continue
    k="grid "+r.get('Grid_Size','?')
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,d in agg.items():
    print(k, {c: f"{v/cnt[(k,c)]:.5g}" for c,v in sorted(d.items())})
PY
done
find $O -name "*counter_collection.csv" -size +2M -delete
