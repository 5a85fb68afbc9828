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
# Round 4 named the instruction form (profiles/r04_hazard.md: the SLP build of sim.hip with single instruction classes rewritten into scalar
# code in the ASSEMBLY, everything else byte-identical): the rollouts differ exactly as long as the kernel holds packed-fp32 ARITHMETIC
# whose LOW result reads the HIGH half of a source register pair (v_pk_mul_f32 / v_pk_add_f32 with op_sel other than [0,0]: 22 of 480
# provoked runs with them, 0 of 672 without; v_pk_mov_b32 and the op_sel_hi-only "broadcast" forms are not involved).  -fno-slp-vectorize
# removes them from every file but embed.hip, whose float4-times-scalar expressions still produced four: that file is compiled without
# packed-fp32 instructions altogether (HBM-bound kernels: no cost).  isa_guard() below refuses an object with such an instruction.
NO_PK = "-Xclang -target-feature -Xclang -packed-fp32-ops"
COMMON = "-fno-slp-vectorize"
SRCS = {"gemm": "", "attention": "", "sim": "-ffp-contract=off", "context": "-ffp-contract=off", "embed": NO_PK,
        "map_encoder": "", "sample": "", "metrics": "-ffp-contract=off", "rewards": "-ffp-contract=off", "forward": "", "dispatch": "", "api": ""}
OUT = os.path.join(HERE, "libctrlsim_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# CTRLSIM_VARIANT=<name> (tools only): a complete second library built with CTRLSIM_EXTRA_DEFS into its own object directory,
# tools/microbench/variants/all_<name>.so (selected at run time with CTRLSIM_LIB=<path>); the product library is untouched
VARIANT = os.environ.get("CTRLSIM_VARIANT", "")
OBJDIR = "build_" + VARIANT if VARIANT else "build"
if VARIANT:
    OUT = os.path.join(HERE, "..", "..", "tools", "microbench", "variants", f"all_{VARIANT}.so")


LLVM_BIN = os.environ.get("CTRLSIM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def jobs():
    """(source name, object name, extra flags) of every object of the library."""
    out = [(name, name, extra) for name, extra in SRCS.items()]
    out += [(name, f"{name}_s{sch}", f"-DCTRLSIM_F16X3={sch}") for name in SPLIT_SRCS for sch in (1, 0)]
    return out


def compile_cmd(name, extra, obj):
    """The shipped command line of one object (tests/test_isa_checks.py compiles with exactly these flags)."""
    src = os.path.join(HERE, name + ".hip")
    return [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj] + COMMON.split() + extra.split()


HOST_ONLY = ("dispatch", "api")            # objects without device code: an EMPTY disassembly is expected there and nowhere else


def _llvm_tool(name):
    path = os.path.join(LLVM_BIN, name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: the ISA guard of the shipped build (csrc/build.py: isa_guard) needs llvm-objcopy, "
                           "clang-offload-bundler and llvm-objdump; point CTRLSIM_LLVM_BIN at the directory that holds them")
    return path


def device_isa(obj):
    """Disassembly (text) of the gfx950 code object embedded in a host object / shared library built by hipcc."""
    import tempfile
    for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"):
        _llvm_tool(t)
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "x.fatbin"), os.path.join(td, "x.co")
        # (objcopy with one file name rewrites that file in place: always name a scratch output)
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(td, "copy.o")],
                           capture_output=True, text=True)
        if r.returncode:                       # a host-only object (dispatch.hip, api.hip) has no device code
            if "not found" in r.stderr:
                return ""
            raise RuntimeError(r.stderr)
        subprocess.check_call([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
        return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout


def packed_counts(isa):
    """(v_pk_mov_b32, packed fp32 arithmetic whose low result reads a high half — op_sel other than all zeros —, packed fp32 arithmetic
    with any operand swizzle, all packed fp32 arithmetic) of a disassembly."""
    import re
    mov = cross = swz = arith = 0
    for ln in isa.splitlines():
        if "v_pk_mov_b32" in ln:
            mov += 1
        elif "v_pk_mul_f32" in ln or "v_pk_add_f32" in ln or "v_pk_fma_f32" in ln:
            arith += 1
            swz += "op_sel" in ln
            m = re.search(r"op_sel:\[([0-9,]+)\]", ln)
            cross += bool(m and "1" in m.group(1))
    return mov, cross, swz, arith


def isa_guard(obj):
    """The build refuses an object that holds packed-fp32 arithmetic with op_sel set (the form the co-residency hazard follows: DESIGN.md
    section 4, profiles/r04_hazard.md) or v_pk_mov_b32 (only the SLP vectoriser emits it: its presence means the pass ran).  A compiler
    upgrade, a new pass or an edit that brings them back must not ship silently."""
    isa = device_isa(obj)
    base = os.path.basename(obj).split(".")[0]
    # a disassembler whose output format changed (or an extraction that silently produced nothing) must not read as "no packed code":
    # every object with device code has to show machine instructions we know are in it
    if base not in HOST_ONLY and ("s_endpgm" not in isa or "v_" not in isa):
        raise RuntimeError(f"{os.path.basename(obj)}: no gfx950 disassembly extracted (llvm tools / bundle format changed?): the ISA guard "
                           "cannot vouch for this object")
    mov, cross, swz, arith = packed_counts(isa)
    if mov or cross:
        raise RuntimeError(f"{os.path.basename(obj)}: {cross} packed-fp32 instructions with op_sel, {mov} v_pk_mov_b32 in the device code "
                           "(co-residency hazard, DESIGN.md section 4): build with -fno-slp-vectorize, or the file without packed fp32 (NO_PK)")
    return swz, arith


def build(force=False, verbose=False):
    deps = [os.path.join(HERE, "common.h"), os.path.join(HERE, "split.h"), os.path.join(HERE, "classes.h"), os.path.join(HERE, "..", "..", "include", "ctrlsim.h")]
    objs = []
    procs = []
    try:
        hipcc_id = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.splitlines()[0:2]
    except OSError:
        hipcc_id = []
    for name, objname, extra in jobs():
        src = os.path.join(HERE, name + ".hip")
        obj = os.path.join(HERE, OBJDIR, objname + ".o")
        os.makedirs(os.path.dirname(obj), exist_ok=True)
        objs.append(obj)
        cmd = compile_cmd(name, extra, obj)
        cmd += os.environ.get("CTRLSIM_EXTRA_DEFS", "").split()   # A/B tuning knobs, e.g. -DGEMM_TBK=16
        # an object is stale when its command line or the compiler changed, too.  The stamp is written AFTER the object exists (a failed
        # or interrupted rebuild must not leave new stamps beside old objects), and hipcc writes to a temporary file that is renamed.
        stamp = obj + ".flags"
        want = " ".join(cmd) + "\n" + "\n".join(hipcc_id)
        same_flags = os.path.exists(stamp) and open(stamp).read() == want
        if force or not same_flags or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            if os.path.exists(stamp):
                os.remove(stamp)
            tmp = obj + ".tmp.o"
            run = cmd[:cmd.index("-o") + 1] + [tmp] + cmd[cmd.index("-o") + 2:]
            if verbose:
                print(" ".join(cmd))
            procs.append((objname, obj, tmp, stamp, want, subprocess.Popen(run)))
    # every compiler process is waited for before anything is raised: a process left running would keep writing its *.tmp.o under the
    # next build's feet.  Compile failures and guard refusals are collected and reported together.
    failed, refused = [], []
    for name, obj, tmp, stamp, want, p in procs:
        if p.wait() != 0:
            failed.append(name)
            continue
        if not VARIANT or os.environ.get("CTRLSIM_ISA_GUARD", "1") != "0":
            try:
                isa_guard(tmp)
            except RuntimeError as e:
                refused.append(str(e))
                os.remove(tmp)
                continue
        os.replace(tmp, obj)
        open(stamp, "w").write(want)
    if failed or refused:
        raise RuntimeError("; ".join((["hipcc failed on " + ", ".join(failed)] if failed else []) + refused))
    if force or procs or not os.path.exists(OUT):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
