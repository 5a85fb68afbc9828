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
