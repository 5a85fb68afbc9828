"""Per-basic-block instruction-class histogram of one kernel in a hipcc -S listing (tuning aid).
usage: python tools/isa_blocks.py file.s <kernel-name-substring> [min_instrs]"""
import collections
import sys


def main():
    path, sub = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l.split(":")[0] and ":" in l)
    blocks, cur = [], ["entry", collections.Counter(), 0]
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if not t or t.startswith(";") or (t.startswith(".") and not t.startswith(".LBB")):
            continue
        if t.startswith(".LBB"):
            blocks.append(cur)
            cur = [t.split(":")[0], collections.Counter(), 0]
            continue
        op = t.split()[0]
        if "mfma" in op: c = "mfma"
        elif op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log")): c = "trans"
        elif op.startswith("v_"): c = "valu"
        elif op.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "s_nop")): c = op
        elif op.startswith("s_"): c = "salu"
        elif op.startswith("ds_"): c = "ds"
        elif op.startswith(("global_", "buffer_", "scratch_")): c = "vmem"
        else: c = op
        cur[1][c] += 1
        cur[2] += 1
    blocks.append(cur)
    for b in blocks:
        if b[2] >= mn:
            print(f"{b[0]:12s} {b[2]:5d}  " + "  ".join(f"{k}:{v}" for k, v in sorted(b[1].items())))


if __name__ == "__main__":
    main()
