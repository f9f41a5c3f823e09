"""Host-side units of the HIP sources that also compile for the CPU (the `HD` functions of csrc/*.hpp): built with
hipcc's host pass and run here, no GPU involved.  The same code runs in the kernels, where tests/test_gpu_parity.py
checks it end to end."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust-eth-kzg_amd", "csrc")


@pytest.mark.timeout(600)
def test_binary_gcd_inversion_matches_fermat(tmp_path):
    """csrc/inverse.hpp (used by the Jacobian -> affine step) against a^(p-2), 3000 values per field incl. edge cases."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "test_inverse")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-I", CSRC,
                           os.path.join(ROOT, "tests", "c", "test_inverse.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatches" in out.stdout
