#!/bin/bash
# SQ counters of the two in_proj kernels (separate --pmc passes, no tracing flags): tools/microbench/pmc_inproj.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pmc_inproj; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $set -d $O/pmc$i -o pmc --output-format csv -- python tools/microbench/pmc_inproj.py > $O/pmc$i.log 2>&1)
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  echo "== set $i"
  python - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    k = "inproj_rs_kernel" if "inproj_rs" in k else ("gemm_ws256_kernel<KV>" if "gemm_ws256" in k else None)
    if not k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k,d in sorted(agg.items()):
    print(k, "dispatches", max(cnt[(k,c)] for c in d), {c: f"{v/cnt[(k,c)]:.4g}" for c,v in sorted(d.items())})
PY
  find $O/pmc$i -name "*counter_collection.csv" -delete
done
