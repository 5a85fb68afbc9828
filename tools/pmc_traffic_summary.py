"""Summarise tools/pmc_traffic.sh output PER KERNEL (round 5): FETCH_SIZE / WRITE_SIZE per launch of every hot kernel of a bench.py run, next to
the algorithmic bytes bench.py itself counted for the same kernels in the same run (`roofline.kernel_bytes_all_streams` of its JSON line).

One counter convention for every file under profiles/ from round 5 on (calibrated on known-byte launches, profiles/r05_pmc_calibration.json):
  HBM bytes = 2 x FETCH_SIZE x 1024 + 1 x WRITE_SIZE x 1024     (gfx950 tallies its 128-byte read requests at 64 B; writes are exact)
usage: pmc_traffic_summary.py <out-dir of pmc_traffic.sh>  ->  JSON on stdout (the format bench.py reads as profiles/r05_pmc_traffic.json)"""
import collections
import csv
import glob
import json
import re
import sys

FETCH_FACTOR, WRITE_FACTOR = 2.0, 1.0


def kind_of(name):
    """kernel name (mangled or demangled) -> bench.py's KIND_KEYS entry, or None for kernels outside the two MFMA classes."""
    if "inproj_rs_kernel" in name:
        return "linear_kv_images"
    if "ffn_fused_bf16x6_kernel" in name:
        # mode 2 = out-projection + LayerNorm + query projection (launch_outproj_ln_q books it as Linear + residual + LayerNorm); modes 0 / 1 =
        # the feed-forward block without / with the out-projection in front of it
        return "linear_residual_layernorm" if re.search(r"ffn_fused_bf16x6_kernel(?:ILi|<)2", name) else "ffn_fused"
    m = re.search(r"attention_bf16x6_kernel(?:ILi|<)(\d)", name)
    if m:
        return "attention_causal" if m.group(1) == "1" else "attention_keypad"
    m = re.search(r"gemm_ws256_kernelILb(\d)ELb(\d)ELb(\d)ELb(\d)E", name) or \
        re.search(r"gemm_nt_bf16x6_kernelILi\dELi\dELi\dELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
    if m:
        relu, resid, ln, kv = (g == "1" for g in m.groups())
        return "linear_kv_images" if kv else "linear_residual_layernorm" if ln else "linear_plain"
    m = re.search(r"gemm_(?:ws256|nt_bf16x6)_kernel<(.*?)>", name)
    if m:
        f = [x.strip() for x in m.group(1).split(",")]
        flags = [x == "true" for x in f[-4:]]
        return "linear_kv_images" if flags[3] else "linear_residual_layernorm" if flags[2] else "linear_plain"
    return None


def per_kernel(path, ctr):
    tot = collections.Counter(); n = collections.Counter()
    for f in glob.glob(f"{path}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"]) * 1024.0
                n[r["Kernel_Name"]] += 1
    return tot, n


def main():
    out = sys.argv[1]
    ft, fn = per_kernel(f"{out}/FETCH_SIZE", "FETCH_SIZE")
    wt, wn = per_kernel(f"{out}/WRITE_SIZE", "WRITE_SIZE")
    line = None
    for ln in open(f"{out}/FETCH_SIZE.log"):
        if ln.startswith("{"):
            line = json.loads(ln)
    # (round 6: the stdout line is the short one; the per-kernel byte counts are in the detail file it names)
    if line and "kernel_bytes_all_streams" not in line.get("roofline", {}) and line.get("detail_file"):
        import os
        for cand in (line["detail_file"], os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", line["detail_file"])):
            if os.path.exists(cand):
                line = json.load(open(cand))
                break
    alg = (line or {}).get("roofline", {}).get("kernel_bytes_all_streams", {})
    kinds = collections.defaultdict(lambda: {"launches": 0, "fetch_counter_bytes": 0.0, "write_counter_bytes": 0.0, "kernel_names": []})
    names = {}
    for k in set(ft) | set(wt):
        kd = kind_of(k)
        short = re.sub(r"\(.*", "", k.replace("void ", ""))[:100]
        names[short] = {"kind": kd, "launches": int(fn.get(k, wn.get(k, 0))),
                        "hbm_bytes_per_launch": (FETCH_FACTOR * ft.get(k, 0.0) / max(fn.get(k, 1), 1) + WRITE_FACTOR * wt.get(k, 0.0) / max(wn.get(k, 1), 1))}
        if kd is None:
            continue
        e = kinds[kd]
        if fn.get(k, 0) != wn.get(k, 0):
            e["launch_count_mismatch"] = [int(fn.get(k, 0)), int(wn.get(k, 0))]
        e["launches"] += int(fn.get(k, 0)); e["fetch_counter_bytes"] += ft.get(k, 0.0); e["write_counter_bytes"] += wt.get(k, 0.0)
        e["kernel_names"].append(short)
    res = {}
    for kd, e in kinds.items():
        n = max(e["launches"], 1)
        hbm = FETCH_FACTOR * e["fetch_counter_bytes"] + WRITE_FACTOR * e["write_counter_bytes"]
        r = {"launches": e["launches"], "fetch_size_counter_bytes_per_launch": e["fetch_counter_bytes"] / n,
             "fetch_bytes_per_launch": FETCH_FACTOR * e["fetch_counter_bytes"] / n, "write_bytes_per_launch": WRITE_FACTOR * e["write_counter_bytes"] / n,
             "hbm_bytes_per_launch": hbm / n, "kernel_names": sorted(e["kernel_names"])}
        a = alg.get(kd)
        if a:
            r["algorithmic_bytes_per_launch_same_run"] = a["algorithmic_hbm_bytes"] / max(a["launches"], 1)
            r["launches_counted_by_bench"] = a["launches"]
            # per LAUNCH on both sides: the counters also see the warm-up steps' launches (same sizes), bench.py counts the timed ones
            r["hbm_over_algorithmic"] = (hbm / n) / (a["algorithmic_hbm_bytes"] / max(a["launches"], 1))
        res[kd] = r
    print(json.dumps({"_convention": f"HBM bytes = {FETCH_FACTOR:g} x FETCH_SIZE + {WRITE_FACTOR:g} x WRITE_SIZE (counter units of 1024 B); factors calibrated on known-byte "
                                     "launches of every access pattern and of each hot kernel: profiles/r05_pmc_calibration.json",
                      "kernels": res, "by_kernel_name": names}, indent=1))


if __name__ == "__main__":
    main()
