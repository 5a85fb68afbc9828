"""Build libctrlsim_hip.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force]

sim.hip and context.hip are compiled with -ffp-contract=off: their float32/float64 arithmetic must equal the
reference's plain IEEE evaluation (no FMA contraction); the MFMA/GEMM files use the default contraction.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# sources -> extra flags.  The three split-operand kernel files are compiled ONCE PER SCHEME (csrc/split.h: -DCTRLSIM_F16X3=1 two fp16
# planes, =0 three bf16 planes; each build lives in its own namespace) and dispatch.hip picks one at run time.
SPLIT_SRCS = ("gemm_bf16x6", "ffn_fused", "attention_bf16x6")
# EVERY source is compiled with -fno-slp-vectorize (COMMON below).  With clang's SLP vectoriser on — it stitches neighbouring scalar
# float operations into packed-fp32 instructions with operand swizzles (v_pk_mul_f32 ... op_sel:[1,0], v_pk_mov_b32) and neighbouring
# LDS accesses into 64 / 128-bit ones — a workgroup that SHARES ITS CU with matrix-pipe workgroups of ANOTHER kernel now and then
# produced different results from identical inputs (the simulator step beside the split-operand GEMM: lanes 48-63 of a wave with a
# wrong x-velocity; few-row matrix kernels beside full-row ones: a flipped token).  Provoked rollouts that differ from the
# single-stream rollout (tools/stress_streams.py, 8 scenes): simulator beside matrix kernels 23/48 with SLP, 0/176 at -O1, 0/112
# with -fno-slp-vectorize, 0/48 with SLP but without packed-fp32 instructions (-target-feature -packed-fp32-ops); few-row kernels on
# the side streams 16/64 with SLP in the matrix kernels, 0/64 without.  No throughput cost (113.39 k vs 113.38 k).
COMMON = "-fno-slp-vectorize"
SRCS = {"gemm": "", "attention": "", "sim": "-ffp-contract=off", "context": "-ffp-contract=off", "embed": "",
        "map_encoder": "", "sample": "", "metrics": "-ffp-contract=off", "rewards": "-ffp-contract=off", "forward": "", "dispatch": "", "api": ""}
OUT = os.path.join(HERE, "libctrlsim_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# CTRLSIM_VARIANT=<name> (tools only): a complete second library built with CTRLSIM_EXTRA_DEFS into its own object directory,
# tools/microbench/variants/all_<name>.so (selected at run time with CTRLSIM_LIB=<path>); the product library is untouched
VARIANT = os.environ.get("CTRLSIM_VARIANT", "")
OBJDIR = "build_" + VARIANT if VARIANT else "build"
if VARIANT:
    OUT = os.path.join(HERE, "..", "..", "tools", "microbench", "variants", f"all_{VARIANT}.so")


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    deps = [os.path.join(HERE, "common.h"), os.path.join(HERE, "split.h"), os.path.join(HERE, "classes.h"), os.path.join(HERE, "..", "..", "include", "ctrlsim.h")]
    objs = []
    procs = []
    jobs = [(name, name, extra) for name, extra in SRCS.items()]
    jobs += [(name, f"{name}_s{sch}", f"-DCTRLSIM_F16X3={sch}") for name in SPLIT_SRCS for sch in (1, 0)]
    for name, objname, extra in jobs:
        src = os.path.join(HERE, name + ".hip")
        obj = os.path.join(HERE, OBJDIR, objname + ".o")
        os.makedirs(os.path.dirname(obj), exist_ok=True)
        objs.append(obj)
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj] + COMMON.split() + extra.split()
        cmd += os.environ.get("CTRLSIM_EXTRA_DEFS", "").split()   # A/B tuning knobs, e.g. -DGEMM_TBK=16
        stamp = obj + ".flags"                                     # an object is stale when its command line changed, too
        same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not same_flags or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            open(stamp, "w").write(" ".join(cmd))
            if verbose:
                print(" ".join(cmd))
            procs.append((objname, subprocess.Popen(cmd)))
    for name, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {name}.hip")
    if force or procs or not os.path.exists(OUT):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
