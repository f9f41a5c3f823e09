"""The drop-in boundary from C: include/c_eth_kzg.h must compile as plain C, and a C program linked against
libc_eth_kzg.so (the way the reference's Go / C# / Nim callers link c_eth_kzg) must reproduce the golden vector."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_runner.c")
LIBDIR = os.path.join(ROOT, "rust-eth-kzg_amd")


def _build(tmp_path):
    exe = str(tmp_path / "abi_runner")
    # link by file path (the reference's consumers use -lc_eth_kzg; the file is libc_eth_kzg.so)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", LIBDIR, "-lc_eth_kzg", "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


def _build_static(tmp_path):
    """The same consumer linked against the static archive, the way bindings/golang/prover.go links libc_eth_kzg.a;
    the HIP runtime and the C++ runtime come from the consumer's link line."""
    exe = str(tmp_path / "abi_runner_static")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                           os.path.join(LIBDIR, "libc_eth_kzg.a"), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                           "-lamdhip64", "-lstdc++", "-ldl", "-lpthread", "-lm", "-o", exe])
    return exe


def test_c_program_links_against_the_static_archive(tmp_path):
    assert os.path.exists(os.path.join(LIBDIR, "libc_eth_kzg.a")), "build the HIP extension first"
    _build_static(tmp_path)


def test_header_is_plain_c():
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                           os.path.join(ROOT, "include", "c_eth_kzg.h")])


def test_c_program_links_against_the_library(tmp_path):
    assert os.path.exists(os.path.join(LIBDIR, "libc_eth_kzg.so")), "build the HIP extension first"
    _build(tmp_path)


@pytest.mark.gpu
def test_c_program_reproduces_the_golden_vector(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import vectors
    case = vectors.load("compute_cells_and_kzg_proofs")["valid_4aedd1a2a3933c3e"]
    comm = vectors.load("blob_to_kzg_commitment")
    exe = _build(tmp_path)
    blob_path, out_path = tmp_path / "blob.bin", tmp_path / "out.bin"
    blob_path.write_bytes(case["input"]["blob"])
    subprocess.check_call([exe, "compute", str(blob_path), str(out_path)])
    out = out_path.read_bytes()
    assert out[:128 * 2048] == b"".join(case["output"][0])
    assert out[128 * 2048:128 * 2048 + 128 * 48] == b"".join(case["output"][1])
    expected_commitment = [c["output"] for c in comm.values() if c["input"]["blob"] == case["input"]["blob"]]
    if expected_commitment:
        assert out[-48:] == expected_commitment[0]
    res = subprocess.run([exe, "verify", str(blob_path)], check=True, capture_output=True, text=True).stdout
    assert "verified=1" in res and "recovered=1" in res
    # the same unchanged C consumer with ETH_KZG_AMD_DEVICES=0,0: its context spans a device list, same bytes
    env = dict(os.environ, ETH_KZG_AMD_DEVICES="0,0", ETH_KZG_AMD_TABLE_GB="22")
    out_list = tmp_path / "out_list.bin"
    subprocess.check_call([exe, "compute", str(blob_path), str(out_list)], env=env)
    assert out_list.read_bytes() == out
    res = subprocess.run([exe, "verify", str(blob_path)], check=True, capture_output=True, text=True, env=env).stdout
    assert "verified=1" in res and "recovered=1" in res
    # the single-process multi-GPU fan-out (one context per visible GPU; one GPU on the test box) from plain C
    res = subprocess.run([exe, "multi", str(blob_path)], check=True, capture_output=True, text=True).stdout
    assert "multi=1" in res
    # the statically linked consumer computes the same bytes
    exe_s = _build_static(tmp_path)
    out2 = tmp_path / "out_static.bin"
    subprocess.check_call([exe_s, "compute", str(blob_path), str(out2)])
    assert out2.read_bytes() == out
    # an invalid blob (all 0xff) must come back as Err with a message, exit code 2
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\xff" * 131072)
    p = subprocess.run([exe, "compute", str(bad), str(out_path)], capture_output=True, text=True)
    assert p.returncode == 2 and "CouldNotDeserializeScalar" in p.stderr
