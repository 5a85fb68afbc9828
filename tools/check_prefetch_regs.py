"""ISA check for csrc/gemm_bf16x6.hip.  The A prefetch is an inline-asm load (tagged "; A-PREFETCH") that hipcc's waitcnt
insertion does not track; the matching counted wait is tagged "; A-WAIT".  Between a tagged load and the next tagged wait no
instruction may touch the destination registers (a register copy or a reuse there would read / clobber data in flight).
Checked on the linear instruction order of every kernel and, for every loop (backward branch) that contains tagged loads, on
two consecutive trips through the loop body.
Usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Ictrl-sim_amd/csrc ctrl-sim_amd/csrc/gemm_bf16x6.hip -o g.s
       python tools/check_prefetch_regs.py g.s"""
import re
import sys

LOAD = re.compile(r'\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], off ; A-PREFETCH')
WAIT = re.compile(r'\s*s_waitcnt vmcnt\(\d+\) ; A-WAIT')


def regs(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def scan(seq, name):
    bad = 0
    for i, l in enumerate(seq):
        m = LOAD.match(l)
        if not m:
            continue
        dst = set(range(int(m.group(1)), int(m.group(2)) + 1))
        for k in seq[i + 1:]:
            if WAIT.match(k):
                break
            ops = k.split(None, 1)
            if len(ops) > 1 and regs(ops[1]) & dst:
                print(name[:70], '| load', l.strip()[:44], '| touched by:', k.strip()[:70])
                bad += 1
                break
    return bad


def main(path):
    txt = open(path).read().split('\n')
    starts = [i for i, l in enumerate(txt) if re.match(r'^_Z.*:\s*(;.*)?$', l)] + [len(txt)]
    bad = loads = loops = 0
    for fi in range(len(starts) - 1):
        lines = txt[starts[fi]:starts[fi + 1]]
        name = lines[0]
        labels = {}
        body = []
        for l in lines:
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m:
                labels[m.group(1)] = len(body)
            elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
                body.append(l)
        loads += sum(1 for l in body if LOAD.match(l))
        bad += scan(body, name)
        for i, l in enumerate(body):
            m = re.match(r'\s*s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                region = body[labels[m.group(1)]:i + 1]
                if any(LOAD.match(x) for x in region):
                    loops += 1
                    bad += scan(region + region, name + ' (loop)')
    print(f'tagged loads: {loads}, loops checked: {loops}, hazards: {bad}')
    return 1 if bad or not loads else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1]))
