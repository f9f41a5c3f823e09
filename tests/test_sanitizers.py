"""AddressSanitizer + UndefinedBehaviorSanitizer on the CPU-compilable code (the GPU pool offers no
sanitizers, and none is needed for this: the oracle is plain C and the host units compile with hipcc's host pass).
  * oracle/*.c driven through every entry point by tests/c/oracle_sanitize_main.c
  * the host units tests/c/test_{inverse,host_mul,linmap,glv,pairing}.cpp (csrc headers compiled for the host; test_curve29 is left
    out: the fully unrolled 14-limb field code takes the instrumenting compiler more than 15 minutes)
  * the host-side concurrency protocols of a context (csrc/host_sync.hpp: the helper pool, the combiner of concurrent verifications,
    pass-slot and lane leases, publish / snapshot / retire of the window tables) under ThreadSanitizer and under ASan + UBSan,
    driven by tests/c/test_host_sync.cpp with fake device passes -- the product (engine.hip, verify_many.hip) uses these very classes
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


def test_knobs_parsing_under_asan_and_ubsan(tmp_path):
    """csrc/knobs.hpp (HIP-free): the device list of eth_kzg_das_context_new, the table budget and the other variables -- well-formed,
    sloppy and garbage values (tests/c/test_knobs.cpp)."""
    exe = str(tmp_path / "test_knobs")
    subprocess.check_call(["g++", "-O1", "-std=c++17", *SAN, "-I", CSRC, os.path.join(ROOT, "tests", "c", "test_knobs.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=ENV)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout + out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_host_concurrency_protocols_under_sanitizers(tmp_path, san):
    """32 threads x mixed operations (combined verifications whose pass sleeps / throws / reports wrong proofs, lane leases, MSM stages
    reading a table that a builder thread is publishing group by group, abandoned tables reaped on the builder's thread, parallel_for
    with a failing index) on two shared contexts while a third party creates and frees contexts: no data race, no lost wake-up, no
    table destroyed on a caller's thread, every verdict right.  The same header builds into libc_eth_kzg.so."""
    exe = str(tmp_path / "host_sync")
    flags = ["-fsanitize=" + san, "-fno-omit-frame-pointer", "-g"] + (["-fno-sanitize-recover=all"] if "address" in san else [])
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", *flags, "-I", CSRC, os.path.join(ROOT, "tests", "c", "test_host_sync.cpp"), "-o", exe])
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    out = subprocess.run([exe, "32", "1500"], capture_output=True, text=True, timeout=800, env=env)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-2000:] + out.stderr[-6000:]
    assert "ThreadSanitizer" not in out.stderr and "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-6000:]
    assert "elsewhere: 0" in out.stdout
    # the product is built from the same header
    assert '#include "host_sync.hpp"' in open(os.path.join(CSRC, "engine.hpp")).read()
