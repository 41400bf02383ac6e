"""Static audit of the gfx950 code object (CPU only: hipcc cross-compiles).  Every kernel's instruction stream
must stay inside the registers its kernel descriptor allocates: a wave that touches a register outside its
allocation corrupts whichever wave the SIMD placed next to it -- silently, only at two or more waves per SIMD.
(One of the hypotheses for the round-1 wrong-result events)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "genie-smem_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_kernels_stay_inside_their_register_allocation(tmp_path):
    asm = str(tmp_path / "kernels.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-x", "hip", os.path.join(CSRC, "kernels.hip"),
                           "--cuda-device-only", "-S", "-o", asm, "-Wno-unused-command-line-argument"])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "reg_audit.py"), asm], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    lines = [l for l in out.stdout.splitlines() if "kernel" in l or "violations" in l]
    assert lines[-1].strip() == "violations: 0"
    # the dominant kernels are present and none of them moves VGPRs into the accumulator half
    body = out.stdout
    for k in ("match_table_kernel", "traverse_kernel", "interval_kernel"):
        assert k in body
