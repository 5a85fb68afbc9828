"""Static checks on generated gfx950 ISA (no GPU needed: hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

import importlib.util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _build_module():
    spec = importlib.util.spec_from_file_location("ctrlsim_build", os.path.join(ROOT, "ctrl-sim_amd", "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _shipped_flags(name):
    """COMMON + the per-file flags of csrc/build.py: the static checks below must look at the code that ships, not at a default -O3 build."""
    b = _build_module()
    return b.COMMON.split() + next(e for n, o, e in b.jobs() if o == name).split()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemm_prefetch_registers_are_untouched_while_in_flight(tmp_path):
    """csrc/gemm_bf16x6.hip prefetches its A rows with inline-asm loads hipcc does not track and waits for them with counted
    vmcnt: between such a load and its wait nothing may read, copy or reuse the destination registers."""
    asm = tmp_path / "gemm.s"
    src = os.path.join(ROOT, "ctrl-sim_amd", "csrc")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + src,
                    os.path.join(src, "gemm_bf16x6.hip"), "-o", str(asm)] + _shipped_flags("gemm_bf16x6_s1"),
                   check=True, capture_output=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_prefetch_regs.py"), str(asm)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "hazards: 0" in r.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_simulator_kernels_use_no_scratch_memory(tmp_path):
    """csrc/sim.hip keeps nothing in scratch (private) memory: its contact code picks polygons, edges and clip outputs at run time,
    and written with local arrays that put 240 bytes per lane there (collide_boxes: see hb_v / clip_segment2)."""
    src = os.path.join(ROOT, "ctrl-sim_amd", "csrc")
    flags = _shipped_flags("sim")
    assert "-ffp-contract=off" in flags and "-fno-slp-vectorize" in flags
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-I" + src, os.path.join(src, "sim.hip"), "-o",
                        str(tmp_path / "sim.o")] + flags, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    sizes = [int(line.rsplit(":", 1)[1].split()[0]) for line in r.stderr.splitlines() if "ScratchSize [bytes/lane]" in line]
    assert len(sizes) >= 2 and all(v == 0 for v in sizes), sizes


# Packed-fp32 instructions per shipped object: (with an operand swizzle, all).  With -fno-slp-vectorize what is left comes from explicit
# ext_vector_type arithmetic in the sources.  The co-residency hazard of DESIGN.md section 4 went away with the SLP vectoriser's code; this
# table is the record of what the shipped ISA holds, so that a compiler upgrade or an edit that changes it is SEEN (update the table
# together with a fresh run of the provocations, tools/jobs/r04_hazard_final.sh).  Round 5: ffn_fused_s1 holds three kernels instead of one
# (feed-forward block; with the out-projection + LayerNorm in front; out-projection + LayerNorm + query projection): 64 / 256 -> 192 / 448,
# all of them op_sel_hi broadcasts of f32x4-times-scalar expressions; provocations re-run on that build (profiles/README.md, round-5 log).
PACKED_BUDGET = {"gemm": (12, 92), "attention": (16, 16), "sim": (15, 83), "embed": (0, 0), "map_encoder": (120, 194),
                 "gemm_bf16x6_s1": (136, 872), "gemm_bf16x6_s0": (128, 864), "ffn_fused_s1": (192, 448), "ffn_fused_s0": (64, 256),
                 "attention_bf16x6_s1": (64, 72), "attention_bf16x6_s0": (64, 72)}
# (round 6: attention_bf16x6_s* holds one more kernel, the resident-keys form of the key-padded kernel: its Q rows x scale_log2e are eight more
#  unswizzled packed multiplies, 64 / 64 -> 64 / 72; tests/test_gpu_hazard.py re-run on that build with the whole GPU suite.
#  map_encoder: the packed-fp32 form of map_pool (explicit f32x2 FMAs, no SLP): 0 / 9 -> 120 / 194, all swizzles op_sel_hi broadcasts of a LOW
#  half — the forms round 4 cleared; provocations of tools/jobs/r06_u.sh on that build)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_shipped_objects_hold_no_packed_moves_and_no_more_packed_arithmetic_than_recorded():
    """Looks at the objects libctrlsim_hip.so is linked from (csrc/build/*.o, built by __graft_entry__.build() with the shipped flags):
    zero packed-fp32 arithmetic with op_sel — the form the co-residency hazard follows (profiles/r04_hazard.md) — and zero v_pk_mov_b32
    anywhere (the build itself refuses them, build.py::isa_guard), and the number of packed-fp32 arithmetic instructions — swizzled
    (op_sel_hi broadcasts) and in total — does not exceed the recorded table."""
    b = _build_module()
    b.build()                                              # incremental: a no-op when __graft_entry__.build() ran
    seen = {}
    for name, objname, extra in b.jobs():
        obj = os.path.join(b.HERE, b.OBJDIR, objname + ".o")
        assert os.path.exists(obj), obj
        assert os.path.getmtime(obj) <= os.path.getmtime(b.OUT) + 1.0, f"{objname}.o is newer than the library"
        stamp = open(obj + ".flags").read()
        assert "-fno-slp-vectorize" in stamp.split("\n")[0].split(), objname
        mov, cross, swz, arith = b.packed_counts(b.device_isa(obj))
        assert mov == 0, f"{objname}: {mov} v_pk_mov_b32"
        assert cross == 0, f"{objname}: {cross} packed-fp32 instructions whose low result reads a high half (op_sel)"
        seen[objname] = (swz, arith)
    over = {k: (v, PACKED_BUDGET.get(k, (0, 0))) for k, v in seen.items()
            if v[0] > PACKED_BUDGET.get(k, (0, 0))[0] or v[1] > PACKED_BUDGET.get(k, (0, 0))[1]}
    assert not over, f"packed-fp32 instruction counts grew (seen, recorded): {over}"


def test_pk_rewrite_tool_serialises_packed_instructions():
    """tools/probes/pk_rewrite.py (the round-4 hazard bisect): register-pair semantics of the rewrites."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
    import pk_rewrite
    src = ["\tv_pk_mov_b32 v[36:37], v[36:37], v[36:37] op_sel:[1,0]\n",
           "\tv_pk_mov_b32 v[68:69], v[70:71], v[68:69] op_sel:[1,0]\n",
           "\tv_pk_mov_b32 v[86:87], v[8:9], v[68:69] op_sel:[1,0]\n",
           "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]\n",
           "\tv_pk_add_f32 v[0:1], v[0:1], v[4:5] neg_lo:[0,1] neg_hi:[0,1]\n",
           "\tv_pk_add_f32 v[2:3], v[2:3], s[4:5] op_sel_hi:[1,0]\n",
           "\tv_add_f32 v1, v2, v3\n"]
    out, st = pk_rewrite.rewrite(src, {"mov", "arith"})
    text = [l.split(";")[0].strip() for l in out]
    assert text[0] == "v_swap_b32 v36, v37"
    assert text[1:3] == ["v_mov_b32 v69, v68", "v_mov_b32 v68, v71"]        # the high half reads the old v68 first
    assert text[3:5] == ["v_mov_b32 v86, v9", "v_mov_b32 v87, v68"]
    assert text[5:7] == ["v_mul_f32_e64 v0, v3, v4", "v_mul_f32_e64 v1, v2, v5"]
    assert text[7:9] == ["v_add_f32_e64 v0, v0, -v4", "v_add_f32_e64 v1, v1, -v5"]
    assert text[9:11] == ["v_add_f32_e64 v2, v2, s4", "v_add_f32_e64 v3, v3, s4"]
    assert text[11] == "v_add_f32 v1, v2, v3" and st["kept"] == 0
