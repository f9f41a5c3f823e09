"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU-compilable code (the GPU pool offers no
sanitizers, and none is needed for this: the oracle is plain C and the host units compile with hipcc's host pass).
  * oracle/*.c driven through every entry point by tests/c/oracle_sanitize_main.c
  * the host units tests/c/test_{inverse,host_mul,linmap,glv,pairing}.cpp (csrc headers compiled for the host; test_curve29 is left
    out: the fully unrolled 14-limb field code takes the instrumenting compiler more than 15 minutes)
A sanitizer report aborts the program (-fno-sanitize-recover): exit code 0 means a clean run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust-eth-kzg_amd", "csrc")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.mark.timeout(900)
def test_oracle_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    src = [os.path.join(ROOT, "oracle", f) for f in ("field.c", "g1.c", "pairing.c", "sha256.c", "kzg.c")]
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-fopenmp", "-Wall", "-Wno-unused-function", *SAN, "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "c", "oracle_sanitize_main.c"), *src, "-o", exe, "-lm"])
    srs = os.path.join(ROOT, "rust-eth-kzg_amd", "data", "trusted_setup_4096.bin")
    out = subprocess.run([exe, srs], capture_output=True, text=True, timeout=800, env=ENV)
    assert out.returncode == 0 and "oracle sanitize run: ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["test_inverse", "test_host_mul", "test_linmap", "test_glv", "test_pairing"])
def test_host_units_under_asan_and_ubsan(tmp_path, name):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / name)
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "-x", "hip", "--cuda-host-only", *SAN, "-I", CSRC,
                           os.path.join(ROOT, "tests", "c", name + ".cpp"), "-o", exe], timeout=600)
    args = [os.path.join(ROOT, "rust-eth-kzg_amd", "data", "trusted_setup_4096.bin")] if name == "test_pairing" else []
    out = subprocess.run([exe, *args], capture_output=True, text=True, timeout=800, env=ENV)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
