"""Unit-level known answers and properties for the CPU oracle, mirroring the reference's in-module unit tests
(SURVEY.md section 4): reduce_bytes_to_scalar_bias edge values (crates/cryptography/bls12_381/src/lib.rs:163-213),
Booth digits (booth_encoding.rs:10-11,56-99), FFT round trips and the G1 FFT against inner products
(polynomial/src/domain.rs:230-308), lincomb against naive sums (lincomb.rs tests)."""
import ctypes as C

import os

import pytest

import oracle_lib
import synth

R, P = synth.R, synth.P
GEN = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
IDENT = b"\xc0" + b"\x00" * 47


def _lib():
    return oracle_lib._lib()


def test_reduce_bytes_to_scalar_edge_cases():
    def red(x):
        out = C.create_string_buffer(32)
        _lib().oracle_reduce_bytes_to_scalar(x.to_bytes(32, "big"), out)
        return int.from_bytes(out.raw, "big")
    assert red(0) == 0 and red(1) == 1
    assert red(R - 1) == R - 1 and red(R) == 0 and red(R + 1) == 1
    two256m1 = 0x1824B159ACC5056F998C4FEFECBC4FF55884B7FA0003480200000001FFFFFFFD
    assert red(2 ** 256 - 1) == two256m1 == (2 ** 256 - 1) % R


def test_booth_digits_reconstruct_the_scalar():
    lib = _lib()
    # the documented table for window size 3 (booth_encoding.rs:10-11): 4-bit slices -> digits
    expected = [0, 1, 1, 2, 2, 3, 3, 4, -4, -3, -3, -2, -2, -1, -1, 0]
    for sl in range(16):
        # place the 4-bit slice so that window 1 (bits 2..5) reads it
        k = sl << 2
        assert lib.oracle_booth_index(1, 3, k.to_bytes(32, "little")) == expected[sl]
    for k in [0, 1, R - 1, R // 3, 2 ** 254 + 12345] + [int.from_bytes(s, "big") for s in synth.seeded_scalars(8, b"booth")]:
        el = k.to_bytes(32, "little")
        for w in (2, 3, 4, 8, 12, 14):
            n = 255 // w + 1
            digits = [lib.oracle_booth_index(i, w, el) for i in range(n)]
            assert all(-(1 << (w - 1)) <= d <= (1 << (w - 1)) for d in digits)
            assert sum(d << (w * i) for i, d in enumerate(digits)) == k


def test_fr_ntt_round_trip_and_definition():
    n = 64
    xs = synth.seeded_scalars(n, b"ntt")
    data = b"".join(xs)
    fwd = oracle_lib.fr_ntt(data, inverse=False)
    assert oracle_lib.fr_ntt(fwd, inverse=True) == data
    # definition: X_k = sum_j x_j w^(jk), w = 7^((r-1)/n)
    w = pow(7, (R - 1) // n, R)
    vals = [int.from_bytes(x, "big") for x in xs]
    for k in (0, 1, 5, 63):
        exp = sum(v * pow(w, j * k, R) for j, v in enumerate(vals)) % R
        assert int.from_bytes(fwd[32 * k:32 * k + 32], "big") == exp
    # coset forms are inverse to each other
    assert oracle_lib.fr_ntt(oracle_lib.fr_ntt(data, coset=1), coset=2) == data


def test_g1_fft_matches_inner_products():
    n = 8
    ks = synth.seeded_scalars(n, b"g1fft")
    pts = [oracle_lib.g1_mul(GEN, k) for k in ks]
    pts[3] = IDENT
    out = oracle_lib.g1_fft(b"".join(pts), inverse=False)
    w = pow(7, (R - 1) // n, R)
    for k in range(n):
        scalars = [pow(w, j * k, R).to_bytes(32, "big") for j in range(n)]
        assert out[48 * k:48 * k + 48] == oracle_lib.g1_msm(b"".join(pts), b"".join(scalars))
    assert oracle_lib.g1_fft(out, inverse=True) == b"".join(pts)


def test_lincomb_against_scalar_arithmetic():
    ks = [int.from_bytes(s, "big") for s in synth.seeded_scalars(5, b"lc")]
    ms = [int.from_bytes(s, "big") for s in synth.seeded_scalars(5, b"lc2")]
    pts = [oracle_lib.g1_mul(GEN, k.to_bytes(32, "big")) for k in ks]
    total = sum(k * m for k, m in zip(ks, ms)) % R
    got = oracle_lib.g1_msm(b"".join(pts), b"".join(m.to_bytes(32, "big") for m in ms))
    assert got == oracle_lib.g1_mul(GEN, total.to_bytes(32, "big"))
    # zero scalars and identity points are skipped (lincomb.rs:12-27); all-zero gives the identity
    assert oracle_lib.g1_msm(b"".join(pts), b"\x00" * 160) == IDENT
    assert oracle_lib.g1_msm(IDENT * 3, b"".join(m.to_bytes(32, "big") for m in ms[:3])) == IDENT
    # -1 * G + G = O
    assert oracle_lib.g1_msm(GEN + GEN, (R - 1).to_bytes(32, "big") + (1).to_bytes(32, "big")) == IDENT


def test_point_codec_edge_cases():
    assert oracle_lib.g1_validate(IDENT) == 0
    assert oracle_lib.g1_validate(b"\xe0" + b"\x00" * 47) != 0          # infinity flag + sign flag
    assert oracle_lib.g1_validate(b"\x40" + b"\x00" * 47) != 0          # compression flag missing
    assert oracle_lib.g1_validate(GEN) == 0
    flipped = bytes([GEN[0] ^ 0x20]) + GEN[1:]                           # the other square root: still a valid point (-G)
    assert oracle_lib.g1_validate(flipped) == 0
    assert oracle_lib.g1_msm(GEN + flipped, (1).to_bytes(32, "big") * 2) == IDENT


def test_adx_multiplication_of_the_timed_build_equals_the_portable_checker(tmp_path):
    """bench.py's cpu_baseline leg times a native build of the oracle with -DORACLE_ADX (Fp products on mulx + two carry chains,
    oracle/field.c: mont_mul6_adx).  The portable __int128 form stays the checker: here the two are compared on 10^6 chained
    pseudo-random pairs inside C (oracle_fp_mul_selftest), and the whole prover of the ADX build must reproduce a golden vector.
    Skipped on a CPU without BMI2 / ADX (the build then falls back to the portable form and says so)."""
    import ctypes
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "liboracle_adx.so")
    src = [os.path.join(root, "oracle", f) for f in ("field.c", "g1.c", "pairing.c", "sha256.c", "kzg.c")]
    subprocess.check_call(["gcc", "-O3", "-march=native", "-DORACLE_ADX", "-fopenmp", "-fPIC", "-std=gnu11", "-shared", "-o", so] + src)
    lib = ctypes.CDLL(so)
    lib.oracle_fp_mul_selftest.restype = ctypes.c_long
    lib.oracle_fp_mul_selftest.argtypes = [ctypes.c_long, ctypes.c_uint64]
    if not lib.oracle_fp_mul_uses_adx():
        pytest.skip("this CPU has no BMI2 / ADX: the native build uses the portable form")
    assert lib.oracle_fp_mul_selftest(1000000, 0x243f6a8885a308d3) == 0
    assert lib.oracle_fp_mul_selftest(1000, 1) == 0
