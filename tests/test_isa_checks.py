"""Static checks on generated gfx950 ISA (no GPU needed: hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_gemm_prefetch_registers_are_untouched_while_in_flight(tmp_path):
    """csrc/gemm_bf16x6.hip prefetches its A rows with inline-asm loads hipcc does not track and waits for them with counted
    vmcnt: between such a load and its wait nothing may read, copy or reuse the destination registers."""
    asm = tmp_path / "gemm.s"
    src = os.path.join(ROOT, "ctrl-sim_amd", "csrc")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + src,
                    os.path.join(src, "gemm_bf16x6.hip"), "-o", str(asm)], check=True, capture_output=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_prefetch_regs.py"), str(asm)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "hazards: 0" in r.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_simulator_kernels_use_no_scratch_memory(tmp_path):
    """csrc/sim.hip keeps nothing in scratch (private) memory: its contact code picks polygons, edges and clip outputs at run time,
    and written with local arrays that put 240 bytes per lane there (collide_boxes: see hb_v / clip_segment2)."""
    src = os.path.join(ROOT, "ctrl-sim_amd", "csrc")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only", "-ffp-contract=off",
                        "-Rpass-analysis=kernel-resource-usage", "-I" + src, os.path.join(src, "sim.hip"), "-o",
                        str(tmp_path / "sim.o")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    sizes = [int(line.rsplit(":", 1)[1].split()[0]) for line in r.stderr.splitlines() if "ScratchSize [bytes/lane]" in line]
    assert len(sizes) >= 2 and all(v == 0 for v in sizes), sizes
