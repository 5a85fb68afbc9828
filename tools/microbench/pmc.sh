#!/bin/bash
# three separate counter passes over the kernel microbench (no tracing flags together with --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set -d $R/gpurun_out/pmc$i -o pmc --output-format csv -- python tools/microbench/pmc_kernels.py > $R/gpurun_out/pmc$i.log 2>&1)
  f=$(find $R/gpurun_out/pmc$i -name "*counter_collection.csv" | head -1)
  echo "== set $i: $f"
  python - "$f" <<'PY'
import csv,sys,collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:48]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    cnt[(k,r['Counter_Name'])]+=1
for k,d in agg.items():
    if "gemm_nt_bf16x6" in k or "attention_bf16x6" in k or "ffn_fused" in k:
        print(k, {c: f"{v/cnt[(k,c)]:.4g}" for c,v in d.items()})
PY
done
