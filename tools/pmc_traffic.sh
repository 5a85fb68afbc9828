#!/bin/bash
# HBM traffic of the hot kernels from the PMC counters, as MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots) prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes (no tracing flags), per KERNEL, FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at 64 B: calibrated,
# profiles/r05_pmc_calibration.json), WRITE_SIZE as counted.
# usage (on the GPU box, from the repo root):  tools/pmc_traffic.sh <out-dir> -- <bench.py command...>
set -e
OUT=$1; shift; shift
mkdir -p "$OUT"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd "$ROOT" && rocprofv3 --pmc $c -d "$OUT/$c" -o pmc --output-format csv -- "$@" > "$OUT/$c.log" 2> "$OUT/$c.err") || true
done
cd "$ROOT"
python tools/pmc_traffic_summary.py "$OUT" > "$OUT/summary.json"
python - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, e in d["kernels"].items():
    print(k, "launches", e["launches"], "hbm/alg", round(e.get("hbm_over_algorithmic", float("nan")), 3), "fetch", round(e["fetch_bytes_per_launch"] / 1e6, 1), "MB write",
          round(e["write_bytes_per_launch"] / 1e6, 1), "MB")
PY
