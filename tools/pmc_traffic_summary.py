"""Summarise tools/pmc_traffic.sh output: per kernel class launches, FETCH_SIZE / WRITE_SIZE totals and per-launch means."""
import collections
import csv
import glob
import json
import sys


def main():
    out = sys.argv[1]
    res = collections.defaultdict(lambda: {"launches": 0})
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(f"{out}/{ctr}/**/*counter_collection.csv", recursive=True)
        if not files:
            continue
        tot = collections.Counter(); n = collections.Counter()
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != ctr:
                continue
            k = r["Kernel_Name"]
            k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
        for k in tot:
            res[k][ctr + "_raw_sum"] = tot[k]
            res[k]["launches"] = n[k]
    # rocprofv3 reports both in KiB-like units of 1024 B? (derived: TCC_EA0_RDREQ*64/1024); keep raw and bytes
    for k, d in res.items():
        f = d.get("FETCH_SIZE_raw_sum"); w = d.get("WRITE_SIZE_raw_sum")
        if f is not None:
            d["fetch_bytes_per_launch_corrected"] = 2.0 * f * 1024.0 / d["launches"]   # x2: gfx950 128-B requests tallied at 64 B
        if w is not None:
            d["write_bytes_per_launch_uncalibrated"] = w * 1024.0 / d["launches"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
