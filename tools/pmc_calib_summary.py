"""Summarise the FETCH_SIZE / WRITE_SIZE calibration passes of tools/jobs/r05_a.sh.
usage: pmc_calib_summary.py <dir with probe_{FETCH,WRITE}_SIZE/, kern_{FETCH,WRITE}_SIZE/ and kern_expect.json, probe_expect.json>
Counter units are 1024 B (rocprofv3's FETCH_SIZE / WRITE_SIZE = request counters x 64 B / 1024).  Prints per kernel the raw bytes the counter
reports per launch, the known bytes, and raw / known — the reciprocal is the factor to apply."""
import collections
import csv
import glob
import json
import sys


def per_kernel(path, ctr):
    tot = collections.Counter(); n = collections.Counter()
    for f in glob.glob(f"{path}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: (tot[k] * 1024.0 / n[k], n[k]) for k in tot}


def main():
    d = sys.argv[1]
    out = {"probes": {}, "kernels": {}}
    pe = json.load(open(f"{d}/probe_expect.json"))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, (b, n) in sorted(per_kernel(f"{d}/probe_{ctr}", ctr).items()):
            name = k.split("(")[0]
            is_rd = name.startswith("rd")
            if (ctr == "FETCH_SIZE") != is_rd:
                continue
            out["probes"][name] = {"counter": ctr, "launches": n, "counter_bytes_per_launch": b, "known_bytes": pe["bytes_per_launch"],
                                   "counter_over_known": b / pe["bytes_per_launch"]}
    ke = json.load(open(f"{d}/kern_expect.json"))["expect"]
    meas = {ctr: per_kernel(f"{d}/kern_{ctr}", ctr) for ctr in ("FETCH_SIZE", "WRITE_SIZE")}
    for label, e in ke.items():
        match = e.get("match", label)
        row = {"expect": e}
        for ctr, key in (("FETCH_SIZE", "read"), ("WRITE_SIZE", "write")):
            hits = [(k, v) for k, v in meas[ctr].items() if match in k]
            if not hits:
                continue
            k, (b, n) = hits[0]
            row[ctr] = {"kernel": k[:120], "launches": n, "counter_bytes_per_launch": b, "known_bytes": e[key], "counter_over_known": b / e[key]}
            if key == "read" and "read_max" in e:
                row[ctr]["counter_over_known_max"] = b / e["read_max"]
        out["kernels"][label] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
