"""Host-side units of the HIP sources that also compile for the CPU (the `HD` functions of csrc/*.hpp): built with
hipcc's host pass and run here, no GPU involved.  The same code runs in the kernels, where tests/test_gpu_parity.py
checks it end to end."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust-eth-kzg_amd", "csrc")


def _build_and_run(tmp_path, name, *args):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / name)
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-x", "hip", "--cuda-host-only", "-I", CSRC,
                           os.path.join(ROOT, "tests", "c", name + ".cpp"), "-o", exe])
    out = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


@pytest.mark.timeout(600)
def test_binary_gcd_inversion_matches_fermat(tmp_path):
    """csrc/inverse.hpp (used by the Jacobian -> affine step) against a^(p-2), 3000 values per field incl. edge cases."""
    out = _build_and_run(tmp_path, "test_inverse")
    assert "0 mismatches" in out


@pytest.mark.timeout(600)
def test_unsaturated_group_law_matches_saturated_formulas(tmp_path):
    """csrc/curve29.hpp + fp29.hpp (14 x 29-bit field, fused a*b + c*d, Z3-based exceptional handling) against the
    plain Jacobian formulas of csrc/curve.hpp: random points, P+P, P-P, identities, mixed chains."""
    out = _build_and_run(tmp_path, "test_curve29")
    assert "0 mismatches" in out


@pytest.mark.timeout(600)
def test_host_multiplication_64_bit_matches_32_bit(tmp_path):
    """csrc/field.hpp: the host's 64-bit-limb Montgomery multiplication against the portable 32-bit CIOS form."""
    out = _build_and_run(tmp_path, "test_host_mul")
    assert "0 mismatches" in out


@pytest.mark.timeout(600)
def test_fk20_proofs_map_compiles_to_its_definition(tmp_path):
    """csrc/g1_linmap.hpp: the two G1 transforms of the prover compiled into a straight-line program of point operations
    (free cyclotomic splits + Karatsuba / Toom-Cook Toeplitz products with the interpolation on the fixed side). Run over
    Fr, the plan and its scheduled slot program must equal IDFT_128 -> keep 64 -> DFT_128; the Hankel building blocks
    are checked for every size and split."""
    out = _build_and_run(tmp_path, "test_linmap")
    assert "0 mismatches" in out and "fk20 plan" in out


@pytest.mark.timeout(600)
def test_unsaturated_fr_matches_saturated_field(tmp_path):
    """csrc/fr29.hpp (9 x 29-bit Fr of the prover's NTT kernels: no carry instructions, Montgomery factor -1) against the
    saturated arithmetic of csrc/field.hpp: products at every bound the kernels use (entry 32 r, 56 r after 12 layers), limb
    re-grouping, the (x << 5) entry from the stored Montgomery form, and the 4096-point Cooley-Tukey network with bit-reversed
    twiddles against the definition of the DFT, without any reduction inside the transform."""
    out = _build_and_run(tmp_path, "test_fr29")
    assert "0 mismatches" in out


@pytest.mark.timeout(600)
def test_glv_split_is_balanced_and_exact(tmp_path):
    """csrc/glv.hpp (the scalar split behind the GLV window table): k1 + k2 lambda == k mod r and |k1|, |k2| <=
    (lambda + 1) / 2 + 1 < 2^127 for 200 k random scalars, small scalars, 0, 1, r - 1, multiples of lambda and the
    branch points of the two balancing steps."""
    out = _build_and_run(tmp_path, "test_glv")
    assert "0 mismatches" in out


@pytest.mark.timeout(600)
def test_host_pairing_building_blocks_and_bilinearity(tmp_path):
    """csrc/host_pairing.cpp: sparse line multiplication, complex and cyclotomic (Granger-Scott) squaring against the
    general Fp12 product on random values; the pairing check on multiples of the trusted setup's [1]_1, [tau]_1 against
    [1]_2, [tau]_2: e(a [tau]_1, [1]_2) e(-a [1]_1, [tau]_2) = 1, the unbalanced pair is not, identities are."""
    out = _build_and_run(tmp_path, "test_pairing", os.path.join(ROOT, "rust-eth-kzg_amd", "data", "trusted_setup_4096.bin"))
    assert "fp12 building blocks: 0 mismatches" in out and "pairing checks: 0 mismatches" in out


@pytest.mark.timeout(900)
def test_signed_13_digit_field_matches_saturated_field_and_python_integers(tmp_path):
    """csrc/fp30.hpp (13 signed 30-bit digits, R = 2^390: the field of the GLV MSM and the constant multiplications) against
    csrc/field.hpp at every operand class (centred, floor digits, lazy differences), the fused product-minus-value forms, the
    one- and two-accumulator product pairs, worst-case digit patterns; then raw digit vectors of 200 rounds against Python's
    integers: out * 2^390 == a * b (mod p), |out| < p, canonical(w) == w mod p."""
    dump = str(tmp_path / "fp30_dump.txt")
    out = _build_and_run(tmp_path, "test_fp30", dump)
    assert "0 mismatches" in out
    p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    R = 1 << 390

    def val(d):
        return sum(x << (30 * i) for i, x in enumerate(d))

    rows = [line.split() for line in open(dump)]
    assert len(rows) == 200 * 11
    for i in range(0, len(rows), 11):
        g = {r[0]: val([int(x) for x in r[1:]]) for r in rows[i:i + 11]}
        a, bu, w, cu, du = g["a"], g["bu"], g["w"], g["cu"], g["du"]
        for name, want, bound in (("mulCU", a * bu, 1), ("mulWC_U", w * a, 1), ("sqr", a * a, 1),
                                  ("sqr_inj2", a * a - (cu + 2 * du) * R, 4), ("mul_add_split", a * w + cu * a, 1)):
            assert (g[name] * R - want) % p == 0 and abs(g[name]) < bound * p, name
        assert g["canon_w"] == w % p


@pytest.mark.timeout(900)
def test_signed_13_digit_group_law_matches_saturated_formulas(tmp_path):
    """csrc/curve30.hpp (XYZZ mixed addition with fused subtractions and floor-digit products, the general addition and doubling of
    the folds, exact slow paths, the packed table entry, conversions to and from the 14 x 29-bit form) against csrc/curve.hpp."""
    out = _build_and_run(tmp_path, "test_curve30")
    assert "0 mismatches" in out
