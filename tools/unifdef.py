"""A small unifdef: resolve `#ifdef / #ifndef / #if NAME / #if !NAME / #else / #endif` blocks of a source file for macros whose state is given,
leave every other conditional untouched.  Used once per round to take the ablation switches of a kernel file out of the shipped source after
the measurements are in (the variants live in the history and in profiles/README.md).
usage: unifdef.py file -DNAME ... -UNAME ...   (rewrites the file in place; a define line `#define NAME ...` of a -D macro is kept)"""
import re
import sys


def main():
    path = sys.argv[1]
    state = {}
    for a in sys.argv[2:]:
        state[a[2:]] = a.startswith("-D")
    out, stack = [], []          # stack entries: [known, emitting_now, parent_emitting, taken_before]
    for ln in open(path).read().split("\n"):
        s = ln.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if)\s+(.*?)\s*(//.*)?$", s)
        emitting = all(e[1] for e in stack)
        if m:
            kind, expr = m.group(1), m.group(2)
            known, val = False, None
            if kind in ("ifdef", "ifndef") and expr in state:
                known, val = True, state[expr] if kind == "ifdef" else not state[expr]
            elif kind == "if":
                mm = re.fullmatch(r"(!?)\s*(?:defined\s*\(\s*(\w+)\s*\)|(\w+))", expr)
                if mm and (mm.group(2) or mm.group(3)) in state:
                    known, val = True, state[mm.group(2) or mm.group(3)] != bool(mm.group(1))
            if known:
                stack.append([True, val, emitting, val])
            else:
                stack.append([False, True, emitting, True])
                if emitting:
                    out.append(ln)
            continue
        if re.match(r"#\s*else\b", s) and stack:
            e = stack[-1]
            if e[0]:
                e[1] = not e[3]
            elif all(x[1] for x in stack[:-1]):
                out.append(ln)
            continue
        if re.match(r"#\s*elif\b", s) and stack:
            e = stack[-1]
            if e[0]:
                raise SystemExit(f"{path}: #elif of a resolved conditional is not handled: {ln}")
            if all(x[1] for x in stack[:-1]):
                out.append(ln)
            continue
        if re.match(r"#\s*endif\b", s) and stack:
            e = stack.pop()
            if not e[0] and all(x[1] for x in stack):
                out.append(ln)
            continue
        if emitting:
            out.append(ln)
    if stack:
        raise SystemExit(f"{path}: unbalanced conditionals")
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
