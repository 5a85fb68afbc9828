#!/bin/bash
# HBM traffic of the MFMA kernel classes from the PMC counters, as MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots)
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no tracing flags), FETCH_SIZE doubled (gfx950
# tallies 128-byte read requests at 64 B), WRITE_SIZE calibrated on a kernel with a known byte count (pass "cal").
# usage (on the GPU box, from the repo root):  tools/pmc_traffic.sh <out-dir> -- <command...>
set -e
OUT=$1; shift; shift
mkdir -p "$OUT"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd "$ROOT" && rocprofv3 --pmc $c -d "$OUT/$c" -o pmc --output-format csv -- "$@" > "$OUT/$c.log" 2>&1) || true
done
cd "$ROOT"
python tools/pmc_traffic_summary.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
