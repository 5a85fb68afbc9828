"""Rewrite packed-fp32 instructions of a gfx950 assembly file into scalar ones (the co-residency hazard).

    python tools/probes/pk_rewrite.py in.s out.s MODE        MODE = mov | swz | arith  (comma-separated for several)

  mov    every v_pk_mov_b32 (dst.lo = src0[op_sel0], dst.hi = src1[op_sel1]) becomes v_mov_b32 / v_swap_b32
  swz    every v_pk_mul_f32 / v_pk_add_f32 that SWIZZLES (an op_sel, or an op_sel_hi other than [1,1]) becomes two VOP3 instructions
  arith  every v_pk_mul_f32 / v_pk_add_f32 with register operands becomes two VOP3 instructions
  cross  only the swizzled ones whose LOW result reads a high half (op_sel other than [0,0])
  bcast  only the swizzled ones with op_sel [0,0] (op_sel_hi other than [1,1]: the high result reads a low half — a broadcast)
  vsrc   only the swizzled ones whose swizzled operand is a VGPR pair (the shipped, clean library swizzles constants / SGPRs in sim.hip)

Instructions the script cannot serialise without a spare register (both halves read a destination register) or whose operand is a
constant read as a 64-bit value are left alone and counted.  The scalar forms are the same IEEE operations (neg modifiers kept), so a
rewritten kernel computes bit-identical results; it exists to find out WHICH packed form is involved in the hazard.
"""
import re
import sys

PAIR = re.compile(r"^([vs])\[(\d+):(\d+)\]$")


def parse_mods(rest):
    mods = {"op_sel": [0, 0], "op_sel_hi": [1, 1], "neg_lo": [0, 0], "neg_hi": [0, 0]}
    for k in list(mods):
        m = re.search(k + r":\[(\d),(\d)\]", rest)
        if m:
            mods[k] = [int(m.group(1)), int(m.group(2))]
    return mods


def half(op, sel):
    """register name of half `sel` of a 64-bit operand, or None (constant / literal: only usable as the low half)"""
    m = PAIR.match(op)
    if m:
        return f"{m.group(1)}{int(m.group(2)) + sel}"
    return None


def rewrite(lines, modes):
    out, stats = [], {"mov": 0, "swz": 0, "arith": 0, "kept": 0}
    for ln in lines:
        s = ln.strip()
        m = re.match(r"^(v_pk_mov_b32|v_pk_mul_f32|v_pk_add_f32)\s+([^,]+),\s*([^,]+),\s*(\S+)(.*)$", s)
        if not m:
            out.append(ln)
            continue
        opc, dst, a, b, rest = m.group(1), m.group(2).strip(), m.group(3).strip(), m.group(4).strip(), m.group(5)
        mods = parse_mods(rest)
        d0, d1 = half(dst, 0), half(dst, 1)
        if opc == "v_pk_mov_b32":
            if "mov" not in modes:
                out.append(ln)
                continue
            x, y = half(a, mods["op_sel"][0]), half(b, mods["op_sel"][1])
            if x is None or y is None:
                stats["kept"] += 1
                out.append(ln)
                continue
            if x == d1 and y == d0:
                new = [f"v_swap_b32 {d0}, {d1}"]
            elif y == d0:
                new = [f"v_mov_b32 {d1}, {y}", f"v_mov_b32 {d0}, {x}"]
            else:
                new = [f"v_mov_b32 {d0}, {x}", f"v_mov_b32 {d1}, {y}"]
            new = [n for n in new if n.split()[1].rstrip(",") != n.split()[2]]      # drop self-moves
            stats["mov"] += 1
            out += ["\t" + n + "\t; was: " + s + "\n" for n in new]
            continue
        swizzled = mods["op_sel"] != [0, 0] or mods["op_sel_hi"] != [1, 1]
        cross = mods["op_sel"] != [0, 0]
        vsrc = any(PAIR.match(op) and op.startswith("v") and (mods["op_sel"][k] != 0 or mods["op_sel_hi"][k] != 1) for k, op in enumerate((a, b)))
        pick = "arith" in modes or ("swz" in modes and swizzled) or ("cross" in modes and cross) or \
            ("bcast" in modes and swizzled and not cross) or ("vsrc" in modes and vsrc)
        if not pick:
            out.append(ln)
            continue
        lo = [half(a, mods["op_sel"][0]), half(b, mods["op_sel"][1])]
        hi = [half(a, mods["op_sel_hi"][0]), half(b, mods["op_sel_hi"][1])]
        # a constant operand: usable where its LOW half is selected (the 32-bit constant itself)
        for k, op in enumerate((a, b)):
            if PAIR.match(op) is None:
                lo[k] = op if mods["op_sel"][k] == 0 else None
                hi[k] = op if mods["op_sel_hi"][k] == 0 else None
        if None in lo or None in hi:
            stats["kept"] += 1
            out.append(ln)
            continue
        base = "v_mul_f32_e64" if opc == "v_pk_mul_f32" else "v_add_f32_e64"
        fmt = lambda d, ops, neg: f"{base} {d}, {'-' if neg[0] else ''}{ops[0]}, {'-' if neg[1] else ''}{ops[1]}"
        lo_i, hi_i = fmt(d0, lo, mods["neg_lo"]), fmt(d1, hi, mods["neg_hi"])
        if d0 in hi and d1 in lo:
            stats["kept"] += 1
            out.append(ln)
            continue
        new = [hi_i, lo_i] if d0 in hi else [lo_i, hi_i]
        stats["swz" if swizzled else "arith"] += 1
        out += ["\t" + n + "\t; was: " + s + "\n" for n in new]
    return out, stats


if __name__ == "__main__":
    src, dst, modes = sys.argv[1], sys.argv[2], set(sys.argv[3].split(","))
    out, stats = rewrite(open(src).readlines(), modes)
    open(dst, "w").writelines(out)
    print(f"{dst}: rewritten mov {stats['mov']}, swizzled {stats['swz']}, plain {stats['arith']}; left packed {stats['kept']}")
