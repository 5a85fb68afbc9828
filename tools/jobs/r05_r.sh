#!/bin/bash
# Round 5, job R: do the 32-byte accumulator-layout stores of the QP kernel reach HBM as 64-byte requests? (TCC_EA_WRREQ vs TCC_EA_WRREQ_64B)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum; do
  (cd $R && timeout 300 rocprofv3 --pmc $c -d $O/$c -o pmc --output-format csv -- python tools/microbench/fusion_proxies.py 64 > $O/$c.log 2> $O/$c.err) || true
done
cd $R
python - <<'PY'
import csv, glob, collections
for c in ("TCC_EA_WRREQ_sum", "TCC_EA_WRREQ_64B_sum"):
    f = glob.glob(f"gpurun_out/r05_r/{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        n = row["Kernel_Name"]
        if "ffn_fused" in n or "ws256" in n or "inproj" in n:
            agg[n[:75]].append(float(row["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(c, k, len(v), "launches, mean", sum(v) / len(v))
PY
