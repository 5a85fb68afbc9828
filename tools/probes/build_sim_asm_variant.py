"""Build a library variant whose csrc/sim.hip is compiled WITH clang's SLP vectoriser (the round-2 flags that showed the
co-residency hazard) and whose device assembly is then edited by tools/probes/pk_rewrite.py before it is assembled.

    python tools/probes/build_sim_asm_variant.py NAME MODE      (MODE: none | mov | swz | arith | mov,swz ...)
        -> tools/microbench/variants/simv_NAME.so   (select with CTRLSIM_LIB=<path>)

How: `hipcc -v -save-temps -c sim.hip` prints the sub-commands of the compilation; the device-side assembler / lld / bundler and the
host-side steps are replayed after the .s file has been rewritten.  The other objects are the shipped ones (csrc/build/*.o).
"""
import os
import shlex
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "ctrl-sim_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def main(name, mode):
    vdir = os.path.join(ROOT, "tools", "microbench", "variants")
    os.makedirs(vdir, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        cmd = [HIPCC, "-v", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-save-temps", "-c",
               os.path.join(CSRC, "sim.hip"), "-o", "sim_v.o"]
        r = subprocess.run(cmd, cwd=td, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-3000:])
        subs = [shlex.split(l.strip()) for l in r.stderr.splitlines() if l.startswith(' "')]
        first_as = next(i for i, c in enumerate(subs) if "-cc1as" in c and "amdgcn-amd-amdhsa" in c)
        s_file = os.path.join(td, "sim-hip-amdgcn-amd-amdhsa-gfx950.s")
        if mode != "none":
            r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probes", "pk_rewrite.py"), s_file, s_file + ".new", mode],
                                capture_output=True, text=True)
            print(r2.stdout.strip(), r2.stderr.strip())
            os.replace(s_file + ".new", s_file)
        for c in subs[first_as:]:
            rr = subprocess.run(c, cwd=td, capture_output=True, text=True)
            if rr.returncode:
                sys.exit("replay failed: " + " ".join(c)[:300] + "\n" + rr.stderr[-3000:])
        obj = os.path.join(vdir, f"sim_{name}.o")
        os.replace(os.path.join(td, "sim_v.o"), obj)
    objs = sorted(os.path.join(CSRC, "build", f) for f in os.listdir(os.path.join(CSRC, "build")) if f.endswith(".o") and f != "sim.o")
    out = os.path.join(vdir, f"simv_{name}.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [obj])
    print("built", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
