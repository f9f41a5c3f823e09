"""GPU parity tests: the HIP path, called through the C ABI (libc_eth_kzg.so), against the CPU oracle
and the reference's golden vectors.  Bit-exact: everything here is integer / byte work.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import importlib
import os

import pytest

import oracle_lib
import synth
import vectors
from conftest import TABLE_BUDGETS, table_budget_ctx
from oracle_lib import OracleError

pytestmark = pytest.mark.gpu
kzg = importlib.import_module("rust-eth-kzg_amd")


@pytest.fixture(scope="module", params=TABLE_BUDGETS, ids=lambda b: f"tables-{b}")
def ctx(request):
    # the torch wheel carries its own HIP runtime: when a test of this module brings torch in, it must have been initialised
    # before the engine's (system) runtime, or it finds no GPU -- same order as bench.py and tests/test_gpu_fullsize.py
    import torch
    torch.cuda.init()
    # every test that takes `ctx` runs on the library's DEFAULT tables (108 GB, mixed 15 / 14-bit windows) and on the widest ones
    yield from table_budget_ctx(request.param, lambda: kzg.DASContext(use_precomp=True))


def _call(fn, *a):
    try:
        return fn(*a)
    except (kzg.KzgError, OracleError):
        return None


# ------------------------------------------------------------------ stage level
def test_field_mul_matches_oracle(ctx):
    import ctypes as C
    lib = kzg.load_library()
    for is_fp, width, mod, ofn in ((1, 48, synth.P, oracle_lib.fp_mul), (0, 32, synth.R, oracle_lib.fr_mul)):
        n = 512
        a = synth.seeded_scalars(n, b"a%d" % is_fp, mod, width)
        b = synth.seeded_scalars(n, b"b%d" % is_fp, mod, width)
        # edge values: 0, 1, mod-1
        a[:3] = [(0).to_bytes(width, "big"), (1).to_bytes(width, "big"), (mod - 1).to_bytes(width, "big")]
        b[:3] = [(mod - 1).to_bytes(width, "big")] * 3
        out = C.create_string_buffer(n * width)
        assert lib.eth_kzg_amd_test_field_mul(ctx.handle, b"".join(a), b"".join(b), out, n, is_fp) == 0
        for i in range(n):
            assert out.raw[i * width:(i + 1) * width] == ofn(a[i], b[i]), (is_fp, i)


def test_fr_ntt4096_matches_oracle(ctx):
    import ctypes as C
    lib = kzg.load_library()
    data = synth.seeded_blob(7)
    brp = lambda v: int(format(v, "012b")[::-1], 2)
    elems = [data[32 * i:32 * i + 32] for i in range(4096)]
    # forward DIF: natural in -> bit-reversed out
    out = C.create_string_buffer(131072)
    assert lib.eth_kzg_amd_test_fr_ntt4096(ctx.handle, data, out, 0) == 0
    exp = oracle_lib.fr_ntt(data, inverse=False)
    got = [out.raw[32 * i:32 * i + 32] for i in range(4096)]
    assert b"".join(got[brp(k)] for k in range(4096)) == exp
    # inverse DIT: bit-reversed in -> natural out (scaled by 1/n)
    inp = b"".join(elems[brp(k)] for k in range(4096))
    assert lib.eth_kzg_amd_test_fr_ntt4096(ctx.handle, inp, out, 1) == 0
    assert out.raw == oracle_lib.fr_ntt(data, inverse=True)


def _some_points(n, with_identity=True):
    """n valid compressed G1 points: multiples of the generator (oracle-computed)."""
    gen = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
                        "6c55e83ff97a1aeffb3af00adb22c6bb")
    sc = synth.seeded_scalars(n, b"pts")
    pts = [oracle_lib.g1_mul(gen, s) for s in sc]
    if with_identity and n > 3:
        pts[2] = b"\xc0" + b"\x00" * 47
    return pts


def test_g1_decompress_matches_oracle(ctx):
    import ctypes as C
    lib = kzg.load_library()
    pts = _some_points(24)
    bad = [
        b"\x00" * 48,                                   # compression flag missing
        b"\xe0" + b"\x00" * 47,                         # infinity with sign bit
        b"\xc0" + b"\x00" * 46 + b"\x01",               # infinity with non-zero x
        b"\x9a\x01\x11\xea\x39\x7f\xe6\x9a\x4b\x1b\xa7\xb6\x43\x4b\xac\xd7\x64\x77\x4b\x84\xf3\x85\x12\xbf"
        b"\x67\x30\xd2\xa0\xf6\xb0\xf6\x24\x1e\xab\xff\xfe\xb1\x53\xff\xff\xb9\xfe\xff\xff\xff\xff\xaa\xab",  # x = p
        b"\x80" + b"\x00" * 46 + b"\x05",               # some small x (on curve or not: oracle decides)
        b"\x80" + b"\x00" * 46 + b"\x04",
    ]
    # points taken from the verify vectors' invalid cases (not on curve / not in subgroup)
    for name, case in vectors.load("verify_cell_kzg_proof_batch").items():
        if name.startswith("invalid_commitment") or name.startswith("invalid_proof"):
            for b in case["input"]["commitments"] + case["input"]["proofs"]:
                if len(b) == 48:
                    bad.append(b)
    # random on-curve points: almost surely OUTSIDE the r-torsion (cofactor ~2^126); the production endomorphism
    # test (mode 1) and the definitional [r]P == O test (mode 2) must both agree with the oracle on them
    import hashlib
    k = 0
    while len(bad) < 160:
        x = int.from_bytes(hashlib.sha512(b"curve-point-%d" % k).digest(), "big") % synth.P
        k += 1
        cand = bytearray(x.to_bytes(48, "big"))
        cand[0] |= 0x80 | (0x20 if k & 1 else 0)
        if oracle_lib.g1_validate(bytes(cand), False) == 0:
            bad.append(bytes(cand))
    allp = pts + bad
    n = len(allp)
    # modes: 0 = no subgroup test, 1 = production (unsaturated field, endomorphism test), 2 = definitional [r]P == O and
    # 3 = endomorphism test on the saturated reference forms
    for check in (0, 1, 2, 3):
        st = (C.c_int32 * n)()
        out = C.create_string_buffer(n * 48)
        assert lib.eth_kzg_amd_test_g1_decompress(ctx.handle, b"".join(allp), n, check, st, out) == 0
        for i, p in enumerate(allp):
            exp = oracle_lib.g1_validate(p, bool(check))
            assert (st[i] != 0) == (exp != 0), (i, p.hex(), st[i], exp)
            if exp == 0:
                assert out.raw[48 * i:48 * i + 48] == p  # decompress -> compress round trip is the identity
    # Up to 8192 points the production test runs with FOUR lanes per point that share its doublings (g1_coop.hpp), above with
    # one lane per point: the list above took the first form, the same points repeated past the limit take the second
    reps = 8192 // n + 2
    st_one = (C.c_int32 * n)()
    out_one = C.create_string_buffer(n * 48)
    assert lib.eth_kzg_amd_test_g1_decompress(ctx.handle, b"".join(allp), n, 1, st_one, out_one) == 0
    st_big = (C.c_int32 * (n * reps))()
    out_big = C.create_string_buffer(n * reps * 48)
    assert lib.eth_kzg_amd_test_g1_decompress(ctx.handle, b"".join(allp) * reps, n * reps, 1, st_big, out_big) == 0
    assert list(st_big) == list(st_one) * reps and out_big.raw == out_one.raw * reps


@pytest.mark.parametrize("inverse", [0, 1])
def test_g1_fft128_matches_oracle(ctx, inverse):
    import ctypes as C
    lib = kzg.load_library()
    lanes = 3
    base = _some_points(128 * lanes)
    # lane 1: second half identity (the (h || 0) shape); lane 2: all equal points (forces P+P and P-P paths)
    ident = b"\xc0" + b"\x00" * 47
    for p in range(64, 128):
        base[128 + p] = ident
    for p in range(128):
        base[256 + p] = base[256]
    inp = b"".join(base)
    out = C.create_string_buffer(len(inp))
    assert lib.eth_kzg_amd_test_g1_fft128(ctx.handle, inp, out, lanes, inverse) == 0
    for l in range(lanes):
        exp = oracle_lib.g1_fft(inp[l * 128 * 48:(l + 1) * 128 * 48], inverse=False) if not inverse else None
        if inverse:
            # oracle inverse includes the 1/128 scaling; the kernel's does not (it is folded into the MSM scalars):
            # compare after scaling the oracle's *input* by 128, i.e. oracle_ifft(128 * x) == kernel_ifft(x)
            s128 = (128).to_bytes(32, "big")
            scaled = b"".join(oracle_lib.g1_mul(inp[(l * 128 + p) * 48:(l * 128 + p + 1) * 48], s128) for p in range(128))
            exp = oracle_lib.g1_fft(scaled, inverse=True)
        assert out.raw[l * 128 * 48:(l + 1) * 128 * 48] == exp, (inverse, l)


# ------------------------------------------------------------------ golden vectors through the C ABI
@pytest.mark.parametrize("name,case", sorted(vectors.load("blob_to_kzg_commitment").items()))
def test_blob_to_kzg_commitment_vectors(ctx, name, case):
    assert _call(ctx.blob_to_kzg_commitment, case["input"]["blob"]) == case["output"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("compute_cells_and_kzg_proofs").items()))
def test_compute_cells_and_kzg_proofs_vectors(ctx, name, case):
    out = _call(ctx.compute_cells_and_kzg_proofs, case["input"]["blob"])
    exp = case["output"]
    if exp is None:
        assert out is None
        assert _call(ctx.compute_cells, case["input"]["blob"]) is None
        return
    assert out is not None
    assert out[0] == exp[0]
    assert out[1] == exp[1]
    assert ctx.compute_cells(case["input"]["blob"]) == exp[0]
    assert b"".join(out[0][:64]) == case["input"]["blob"]


def test_compute_cells_and_kzg_proofs_seeded_batch_vs_oracle(ctx, oracle):
    """Synthetic seeded blobs + the reference's dummy blob, as one ragged batch (n = 5, not a multiple of 64)."""
    blobs = [synth.seeded_blob(i) for i in range(4)] + [synth.dummy_blob()]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    assert st == [0] * len(blobs)
    for b, blob in enumerate(blobs):
        ec, ep = oracle.compute_cells_and_kzg_proofs(blob)
        assert cells[b] == ec and proofs[b] == ep, b
    st2, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    assert st2 == [0] * len(blobs)
    for b, blob in enumerate(blobs):
        assert comms[b] == oracle.blob_to_kzg_commitment(blob)


def test_batch_with_invalid_blob_reports_per_blob_status(ctx, oracle):
    good = synth.seeded_blob(11)
    bad = bytearray(good)
    bad[32 * 2111:32 * 2112] = synth.R.to_bytes(32, "big")  # element == r exactly (vector 9d88c338's shape)
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch([good, bytes(bad), good])
    assert st[0] == 0 and st[2] == 0 and st[1] != 0
    ec, ep = oracle.compute_cells_and_kzg_proofs(good)
    assert cells[0] == ec and proofs[0] == ep and cells[2] == ec and proofs[2] == ep


def test_use_precomp_false_gives_identical_results(oracle):
    c = kzg.DASContext(use_precomp=False)
    try:
        blob = synth.seeded_blob(3)
        assert c.compute_cells_and_kzg_proofs(blob) == tuple(oracle.compute_cells_and_kzg_proofs(blob))
        assert c.blob_to_kzg_commitment(blob) == oracle.blob_to_kzg_commitment(blob)
    finally:
        c.close()


# ------------------------------------------------------------------ verify / recover
@pytest.mark.parametrize("name,case", sorted(vectors.load("verify_cell_kzg_proof_batch").items()))
def test_verify_cell_kzg_proof_batch_vectors(ctx, name, case):
    i = case["input"]
    out = _call(ctx.verify_cell_kzg_proof_batch, i["commitments"], i["cell_indices"], i["cells"], i["proofs"])
    assert out == case["output"]


def _verify_in_shards(ctx, world, C_, I_, L_, P_):
    """What `world` ranks would do, run one after another on this GPU: a partial per slice, then the combine."""
    sh = importlib.import_module("rust-eth-kzg_amd.sharding")
    parts = [ctx.verify_cell_kzg_proof_batch_partial(C_, I_, L_, P_, *sh.shard_bounds(len(L_), world, r)) for r in range(world)]
    return ctx.verify_cell_kzg_proof_batch_combine(parts)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("name,case", sorted(vectors.load("verify_cell_kzg_proof_batch").items()))
def test_verify_cell_kzg_proof_batch_vectors_sharded(ctx, name, case, world):
    """The reference's verify vectors through the multi-GPU form: true / false / error must
    come out exactly as in the unsharded call, for every way of slicing the cell list."""
    i = case["input"]
    out = _call(_verify_in_shards, ctx, world, i["commitments"], i["cell_indices"], i["cells"], i["proofs"])
    assert out == case["output"]


def test_verify_sharded_partials_are_additive(ctx):
    """3 blobs x 128 cells: uneven slices, an empty slice and a single-cell slice all combine to the unsharded verdict;
    a tampered cell flips the verdict whichever slice holds it."""
    blobs = [synth.seeded_blob(20 + i) for i in range(3)]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    C_, I_, L_, P_ = [], [], [], []
    for b in range(3):
        for k in range(128):
            C_.append(comms[b]); I_.append(k); L_.append(cells[b][k]); P_.append(proofs[b][k])
    cuts = [0, 1, 1, 200, 384]
    parts = [ctx.verify_cell_kzg_proof_batch_partial(C_, I_, L_, P_, a, b) for a, b in zip(cuts, cuts[1:])]
    assert parts[1] == (b"\xc0" + bytes(47)) * 2  # the empty slice contributes the identity twice
    assert ctx.verify_cell_kzg_proof_batch_combine(parts) is True
    L2 = list(L_); L2[300] = cells[0][1]
    assert _verify_in_shards(ctx, 4, C_, I_, L2, P_) is False
    bad = [ctx.verify_cell_kzg_proof_batch_partial(C_, I_, L2, P_, a, b) for a, b in zip(cuts, cuts[1:])]
    assert ctx.verify_cell_kzg_proof_batch_combine(bad) is False
    # every slice is a weighted sum of per-cell equations, so the slices without the tampered cell still balance
    assert ctx.verify_cell_kzg_proof_batch_combine(bad[:-1]) is True
    assert _verify_in_shards(ctx, 1, C_, I_, L_, P_) is True
    with pytest.raises(kzg.KzgError):
        ctx.verify_cell_kzg_proof_batch_partial(C_, I_, L_, P_, 5, 385)


def test_verify_with_byte_shifted_lincombs(ctx, monkeypatch):
    """Large batches run the two lincombs over byte-shifted point copies built before the challenge is known
    (k_verify.hip: k_pip_shift, k_ps_*).  Forced on for every size here: all of the reference's verify vectors, unsharded
    and in three slices, and the two pairing inputs of a 384-cell batch byte for byte against the windowed form."""
    monkeypatch.setenv("ETH_KZG_AMD_PIP_SHIFT_MIN", "1")
    c = kzg.DASContext()
    try:
        for name, case in sorted(vectors.load("verify_cell_kzg_proof_batch").items()):
            i = case["input"]
            args = (i["commitments"], i["cell_indices"], i["cells"], i["proofs"])
            assert _call(c.verify_cell_kzg_proof_batch, *args) == case["output"], name
            assert _call(_verify_in_shards, c, 3, *args) == case["output"], name
        blobs = [synth.seeded_blob(20 + i) for i in range(3)]
        st, cells, proofs = c.compute_cells_and_kzg_proofs_batch(blobs)
        _, comms = c.blob_to_kzg_commitment_batch(blobs)
        C_, I_, L_, P_ = [], [], [], []
        for b in range(3):
            for k in range(128):
                C_.append(comms[b]); I_.append(k); L_.append(cells[b][k]); P_.append(proofs[b][k])
        # identity proofs / commitments (the zero polynomial) ride along: the point (0,0) is its own shifted copy
        zc, zp = c.compute_cells_and_kzg_proofs(bytes(131072))
        for k in (0, 77):
            C_.append(b"\xc0" + bytes(47)); I_.append(k); L_.append(zc[k]); P_.append(zp[k])
        for lo, hi in [(0, len(L_)), (0, 1), (5, 200), (384, 386)]:
            assert c.verify_cell_kzg_proof_batch_partial(C_, I_, L_, P_, lo, hi) == \
                ctx.verify_cell_kzg_proof_batch_partial(C_, I_, L_, P_, lo, hi), (lo, hi)
        assert c.verify_cell_kzg_proof_batch(C_, I_, L_, P_) is True
        L2 = list(L_); L2[300] = cells[0][1]
        assert c.verify_cell_kzg_proof_batch(C_, I_, L2, P_) is False
        P2 = list(P_); P2[5] = proofs[1][7]
        assert c.verify_cell_kzg_proof_batch(C_, I_, L_, P2) is False
    finally:
        c.close()


@pytest.mark.parametrize("name,case", sorted(vectors.load("recover_cells_and_kzg_proofs").items()))
def test_recover_cells_and_kzg_proofs_vectors(ctx, name, case):
    i = case["input"]
    out = _call(ctx.recover_cells_and_kzg_proofs, i["cell_indices"], i["cells"])
    exp = case["output"]
    if exp is None:
        assert out is None
    else:
        assert out is not None and out[0] == exp[0] and out[1] == exp[1]


def test_verify_multi_blob_batch_and_tamper(ctx):
    """BASELINE.json config 3 in miniature: several blobs x 128 cells in one call; then one tampered proof,
    one tampered cell and a swapped commitment must each flip the result to False (not an error)."""
    blobs = [synth.seeded_blob(20 + i) for i in range(3)]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    C, I, L, P = [], [], [], []
    for b in range(len(blobs)):
        for k in range(128):
            C.append(comms[b]); I.append(k); L.append(cells[b][k]); P.append(proofs[b][k])
    assert ctx.verify_cell_kzg_proof_batch(C, I, L, P) is True
    P2 = list(P); P2[5] = proofs[1][7]
    assert ctx.verify_cell_kzg_proof_batch(C, I, L, P2) is False
    L2 = list(L); L2[300] = cells[0][1]
    assert ctx.verify_cell_kzg_proof_batch(C, I, L2, P) is False
    C2 = list(C); C2[0] = comms[2]
    assert ctx.verify_cell_kzg_proof_batch(C2, I, L, P) is False
    # ragged subset, arbitrary order, duplicates
    sel = [3, 130, 131, 3, 383, 256]
    assert ctx.verify_cell_kzg_proof_batch([C[s] for s in sel], [I[s] for s in sel], [L[s] for s in sel], [P[s] for s in sel]) is True


@pytest.mark.parametrize("pattern", ["even", "first_half", "second_half", "random65", "all"])
def test_recover_round_trip(ctx, oracle, pattern):
    """encode -> erase -> decode round trip (BASELINE.json config 5 shape), checked against the oracle too."""
    import random
    blob = synth.seeded_blob(40)
    cells, proofs = ctx.compute_cells_and_kzg_proofs(blob)
    idx = {"even": list(range(0, 128, 2)), "first_half": list(range(64)), "second_half": list(range(64, 128)),
           "random65": sorted(random.Random(1).sample(range(128), 65)), "all": list(range(128))}[pattern]
    rc, rp = ctx.recover_cells_and_kzg_proofs(idx, [cells[i] for i in idx])
    assert rc == cells and rp == proofs
    oc, op = oracle.recover_cells_and_kzg_proofs(idx, [cells[i] for i in idx])
    assert rc == oc and rp == op
    # inconsistent data (a cell from another blob) must be rejected: recovered polynomial has degree >= 4096
    other, _ = ctx.compute_cells_and_kzg_proofs(synth.seeded_blob(41))
    if pattern != "all" and len(idx) > 64:
        bad = [cells[i] for i in idx]
        bad[0] = other[idx[0]]
        assert _call(ctx.recover_cells_and_kzg_proofs, idx, bad) is None


def test_recover_batch_ragged(ctx, oracle):
    """Batched recovery: different erasure patterns per blob, one inconsistent and one invalid entry in the batch."""
    import random
    blobs = [synth.seeded_blob(60 + i) for i in range(4)]
    _, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    pats = [list(range(0, 128, 2)), list(range(64, 128)), sorted(random.Random(5).sample(range(128), 90)), list(range(128))]
    batch = [(pats[b], [cells[b][i] for i in pats[b]]) for b in range(4)]
    bad_cells = [cells[0][i] for i in pats[2]]
    bad_cells[3] = cells[1][pats[2][3]]
    batch.append((pats[2], bad_cells))                      # inconsistent -> status 4
    batch.append(([1, 0] + list(range(2, 70)), [cells[0][i] for i in [1, 0] + list(range(2, 70))]))  # not ascending -> 3
    st, rc, rp = ctx.recover_cells_and_kzg_proofs_batch(batch)
    assert st[:4] == [0, 0, 0, 0] and st[4] == 4 and st[5] == 3
    for b in range(4):
        assert rc[b] == cells[b] and rp[b] == proofs[b]


# ------------------------------------------------------------------ EIP-4844 single-point operations
@pytest.mark.parametrize("name,case", sorted(vectors.load("compute_kzg_proof").items()))
def test_compute_kzg_proof_vectors(ctx, name, case):
    i = case["input"]
    out = _call(ctx.compute_kzg_proof, i["blob"], i["z"])
    assert (list(out) if out is not None else None) == case["output"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("compute_blob_kzg_proof").items()))
def test_compute_blob_kzg_proof_vectors(ctx, name, case):
    i = case["input"]
    assert _call(ctx.compute_blob_kzg_proof, i["blob"], i["commitment"]) == case["output"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("verify_kzg_proof").items()))
def test_verify_kzg_proof_vectors(ctx, name, case):
    i = case["input"]
    assert _call(ctx.verify_kzg_proof, i["commitment"], i["z"], i["y"], i["proof"]) == case["output"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("verify_blob_kzg_proof").items()))
def test_verify_blob_kzg_proof_vectors(ctx, name, case):
    i = case["input"]
    assert _call(ctx.verify_blob_kzg_proof, i["blob"], i["commitment"], i["proof"]) == case["output"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("verify_blob_kzg_proof_batch").items()))
def test_verify_blob_kzg_proof_batch_vectors(ctx, name, case):
    i = case["input"]
    assert _call(ctx.verify_blob_kzg_proof_batch, i["blobs"], i["commitments"], i["proofs"]) == case["output"]


def test_eip4844_round_trip_on_synthetic_blobs(ctx, oracle):
    blobs = [synth.seeded_blob(80 + i) for i in range(3)]
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    proofs = [ctx.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, comms)]
    assert proofs == [oracle.compute_blob_kzg_proof(b, c) for b, c in zip(blobs, comms)]
    assert ctx.verify_blob_kzg_proof_batch(blobs, comms, proofs) is True
    assert ctx.verify_blob_kzg_proof_batch(blobs, comms, proofs[::-1]) is False
    z = synth.seeded_scalars(1, b"z4844")[0]
    p, y = ctx.compute_kzg_proof(blobs[0], z)
    assert (p, y) == oracle.compute_kzg_proof(blobs[0], z)
    assert ctx.verify_kzg_proof(comms[0], z, y, p) is True
    assert ctx.verify_kzg_proof(comms[1], z, y, p) is False


@pytest.mark.parametrize("n", [113, 114, 130, 143])
def test_commitment_batches_around_one_round_of_wave_slots(ctx, oracle, n):
    """The MSM stage runs an overflow of at most a quarter round of wave slots as its own launch (engine_prover.hip: launch_msm_range);
    for commitments (64 groups per blob) that is 114 .. 142 blobs on nine windows, 129 .. 160 on eight.  Every commitment of the batch
    must equal the single call's, and the oracle's for a sample."""
    import numpy as np
    rng = np.random.RandomState(2000 + n)
    a = rng.randint(0, 256, size=(n, 4096, 32), dtype=np.uint8)
    a[:, :, 0] &= 0x3F
    blobs = [a[i].tobytes() for i in range(n)]
    blobs[n - 1] = synth.dummy_blob()
    st, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    assert st == [0] * n
    for b in sorted({0, 1, n // 2, n - 14, n - 2, n - 1}):
        assert comms[b] == ctx.blob_to_kzg_commitment(blobs[b]), (n, b)
    for b in (0, n - 1):
        assert comms[b] == oracle.blob_to_kzg_commitment(blobs[b]), (n, b)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 8, 9, 12, 13, 16, 17, 32, 33, 64, 65, 128, 129, 192, 193, 200, 256, 257, 320, 321])
def test_every_batch_size_regime_matches_oracle(ctx, oracle, n):
    """The engine picks its G1 schedule by batch size (<= 2: flat MSM over 4 scalar segments + segmented
    doubling chains + circulant transforms, above: the compiled linear map (flat MSM up to 8 blobs) -- in the compilation that suits the
    number of 64-blob lane groups: 712 constant multiplications for one group, 456 for two, 606 for three, 456 for four, 372 for
    five, the 350 of the operation-count optimum from six; inside one lane group the map's constant multiplications take four
    lanes per blob up to 16 blobs, two from 17 to 64 -- and from 33 the 456-multiplication compilation, two waves each -- and its
    cheap levels four (g1_coop.hpp); one blob takes the four-lanes-per-chain circulant kernels;
    MSM: flat <= 8 (and for a small overflow beyond one round of wave slots), windowed below 256 blobs, chunked above).  Every regime and both sides of every threshold
    must give the oracle's bytes; blobs not checked against the oracle are checked against the single-blob path."""
    import numpy as np
    rng = np.random.RandomState(1000 + n)
    a = rng.randint(0, 256, size=(n, 4096, 32), dtype=np.uint8)
    a[:, :, 0] &= 0x3F
    blobs = [a[i].tobytes() for i in range(n)]
    blobs[n // 2] = synth.dummy_blob()
    blobs[0] = bytes(131072)                                  # zero polynomial: every G1 intermediate is the identity
    blobs[n - 1] = (b"\x00" * 31 + b"\x07") * 4096           # constant polynomial: identity proofs, non-trivial commitment
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    assert st == [0] * n
    assert proofs[0] == [b"\xc0" + bytes(47)] * 128 and proofs[n - 1] == [b"\xc0" + bytes(47)] * 128
    for b in sorted({0, n // 2, n - 1}):
        ec, ep = oracle.compute_cells_and_kzg_proofs(blobs[b])
        assert cells[b] == ec and proofs[b] == ep, (n, b)
    for b in sorted({1 % n, n // 3, (2 * n) // 3}):
        c1, p1 = ctx.compute_cells_and_kzg_proofs(blobs[b])
        assert cells[b] == c1 and proofs[b] == p1, (n, b)
    st2, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    assert st2 == [0] * n and comms[n - 1] == oracle.blob_to_kzg_commitment(blobs[n - 1])


def test_batch_larger_than_one_lane_group(ctx, oracle):
    """70 blobs: more than one 64-lane group, with a duplicate blob and the
    all-(r-1) fixture inside. Spot-check against the oracle and check the data-in-first-half invariant on all."""
    import numpy as np
    rng = np.random.RandomState(123)
    a = rng.randint(0, 256, size=(70, 4096, 32), dtype=np.uint8)
    a[:, :, 0] &= 0x3F
    blobs = [a[i].tobytes() for i in range(70)]
    blobs[1] = blobs[0]                                   # duplicate blob
    blobs[65] = vectors.load("compute_cells_and_kzg_proofs")["valid_419245fbfe69f145"]["input"]["blob"]  # all r-1: identity proofs
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    assert st == [0] * 70
    for b in range(70):
        assert b"".join(cells[b][:64]) == blobs[b]
    assert cells[0] == cells[1] and proofs[0] == proofs[1]
    for b in (0, 63, 64, 65, 69):
        ec, ep = oracle.compute_cells_and_kzg_proofs(blobs[b])
        assert cells[b] == ec and proofs[b] == ep, b


def test_golden_vectors_through_the_batch_entry_points(ctx):
    """The reference's vectors again, but as ONE batch per family through the eth_kzg_amd_*_batch entry points: valid
    cases must reproduce the golden bytes, invalid ones must be flagged in their own status slot without disturbing
    their neighbours (cases whose buffers have the wrong length cannot be expressed in the batch ABI and are skipped)."""
    cases = [c for _, c in sorted(vectors.load("compute_cells_and_kzg_proofs").items()) if len(c["input"]["blob"]) == 131072]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch([c["input"]["blob"] for c in cases])
    assert sum(1 for c in cases if c["output"] is None) >= 2 and sum(1 for c in cases if c["output"]) == 7
    for b, c in enumerate(cases):
        if c["output"] is None:
            assert st[b] != 0
        else:
            assert st[b] == 0 and cells[b] == c["output"][0] and proofs[b] == c["output"][1]
    cases = [c for _, c in sorted(vectors.load("blob_to_kzg_commitment").items()) if len(c["input"]["blob"]) == 131072]
    st, comms = ctx.blob_to_kzg_commitment_batch([c["input"]["blob"] for c in cases])
    for b, c in enumerate(cases):
        assert (st[b] != 0) if c["output"] is None else (st[b] == 0 and comms[b] == c["output"])
    cases = [c for _, c in sorted(vectors.load("recover_cells_and_kzg_proofs").items())
             if all(len(x) == 2048 for x in c["input"]["cells"])]
    st, rc, rp = ctx.recover_cells_and_kzg_proofs_batch([(c["input"]["cell_indices"], c["input"]["cells"]) for c in cases])
    assert sum(1 for c in cases if c["output"]) == 4
    for b, c in enumerate(cases):
        if c["output"] is None:
            assert st[b] != 0, b
        else:
            assert st[b] == 0 and rc[b] == c["output"][0] and rp[b] == c["output"][1]


def test_recover_device_resident(ctx):
    """eth_kzg_amd_recover_cells_and_proofs_device: extended blobs in HBM with cells knocked out (their bytes overwritten
    with junk - missing cells must never be read), one presence mask per blob; every erasure pattern of the host API's
    tests, a blob with too few cells and a blob with a non-canonical element, all in one call."""
    import torch
    import numpy as np
    n = 6
    blobs = [synth.seeded_blob(300 + i) for i in range(n)]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    rng = np.random.RandomState(5)
    patterns = [list(range(0, 128, 2)), list(range(64)), list(range(64, 128)), sorted(rng.choice(128, 65, replace=False).tolist()),
                list(range(128)), list(range(63))]
    flat = bytearray()
    for b in range(n):
        for c in range(128):
            flat += cells[b][c] if c in patterns[b] else b"\xff" * 2048  # junk (not even canonical) where the cell is missing
    d_in = torch.frombuffer(flat, dtype=torch.uint8).cuda()
    d_cells = torch.empty(n * 128 * 2048, dtype=torch.uint8, device="cuda")
    d_proofs = torch.empty(n * 128 * 48, dtype=torch.uint8, device="cuda")
    status = ctx.recover_cells_and_kzg_proofs_device(n, d_in.data_ptr(), patterns, d_cells.data_ptr(), d_proofs.data_ptr())
    torch.cuda.synchronize()
    assert status[:5] == [0] * 5 and status[5] == 3
    out_c, out_p = bytes(d_cells.cpu().numpy()), bytes(d_proofs.cpu().numpy())
    for b in range(5):
        assert out_c[b * 128 * 2048:(b + 1) * 128 * 2048] == b"".join(cells[b]), b
        assert out_p[b * 128 * 48:(b + 1) * 128 * 48] == b"".join(proofs[b]), b
    # a present cell with an element >= r is reported for its blob only
    bad = bytearray(flat)
    bad[(1 * 128 + 3) * 2048:(1 * 128 + 3) * 2048 + 32] = synth.R.to_bytes(32, "big")
    d_bad = torch.frombuffer(bad, dtype=torch.uint8).cuda()
    status = ctx.recover_cells_and_kzg_proofs_device(n, d_bad.data_ptr(), patterns, d_cells.data_ptr(), d_proofs.data_ptr())
    assert status == [0, 1, 0, 0, 0, 3]
    # cells that are not evaluations of one degree < 4096 polynomial (needs redundancy: blob 4 has all 128 cells;
    # exactly 64 cells, as in blob 2, always interpolate)
    mixed = bytearray(flat)
    mixed[(4 * 128 + 70) * 2048:(4 * 128 + 71) * 2048] = cells[0][70]
    mixed[(2 * 128 + 70) * 2048:(2 * 128 + 71) * 2048] = cells[0][70]
    d_mix = torch.frombuffer(mixed, dtype=torch.uint8).cuda()
    status = ctx.recover_cells_and_kzg_proofs_device(n, d_mix.data_ptr(), patterns, d_cells.data_ptr(), d_proofs.data_ptr())
    assert status == [0, 0, 0, 0, 4, 3]


def test_second_context_shares_the_window_tables(ctx, oracle):
    """Several contexts in one process (the reference's Java test creates them freely): the second one must come up
    without another 249 GB of tables - it shares the first one's - and give identical results."""
    import torch
    free_before = torch.cuda.mem_get_info()[0]
    c2 = kzg.DASContext(use_precomp=True)
    try:
        assert c2.window_bits() == ctx.window_bits() and c2.table_bytes() == ctx.table_bytes()
        # shared, not rebuilt: the device's free memory must not have dropped by anything like a table (no clock involved)
        used = free_before - torch.cuda.mem_get_info()[0]
        assert used < 3e9, f"second context took {used / 1e9:.1f} GB of HBM: tables were rebuilt"
        blob = synth.seeded_blob(90)
        assert c2.compute_cells_and_kzg_proofs(blob) == ctx.compute_cells_and_kzg_proofs(blob)
    finally:
        c2.close()
    assert ctx.blob_to_kzg_commitment(synth.seeded_blob(91)) == oracle.blob_to_kzg_commitment(synth.seeded_blob(91))


def test_narrower_window_table_gives_identical_results(ctx, oracle, monkeypatch):
    """use_precomp = false next to the default context: the narrowest tables there are (sixteen windows, the last step of the
    ladder when HBM is short) change nothing but speed: same cells, proofs and commitment.  (Every GLV width and every
    budget: tests/test_gpu_tables.py, in processes that do not hold the widest tables already -- a live table is shared.)"""
    c2 = kzg.DASContext(use_precomp=False)
    try:
        assert c2.window_bits() == 8 and c2.table_bytes() < ctx.table_bytes()
        blobs = [synth.seeded_blob(92 + i) for i in range(3)]
        assert c2.compute_cells_and_kzg_proofs_batch(blobs) == ctx.compute_cells_and_kzg_proofs_batch(blobs)
        assert c2.compute_cells_and_kzg_proofs(blobs[0]) == tuple(oracle.compute_cells_and_kzg_proofs(blobs[0]))
    finally:
        c2.close()


def test_context_is_usable_from_many_threads(ctx, oracle):
    """The reference's context is shared across threads (node async_* methods, Java tests): concurrent calls on one
    context must all return correct results (calls serialise inside the library)."""
    import concurrent.futures as cf
    blobs = [synth.seeded_blob(200 + i) for i in range(6)]
    expected = [oracle.blob_to_kzg_commitment(b) for b in blobs]
    cells0, proofs0 = ctx.compute_cells_and_kzg_proofs(blobs[0])

    def work(i):
        if i % 2 == 0:
            return ("c", i, ctx.blob_to_kzg_commitment(blobs[i % 6]))
        c, p = ctx.compute_cells_and_kzg_proofs(blobs[0])
        return ("p", i, (c, p))

    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        for kind, i, out in ex.map(work, range(24)):
            if kind == "c":
                assert out == expected[i % 6]
            else:
                assert out == (cells0, proofs0)


def test_empty_and_degenerate_batches(ctx):
    assert ctx.compute_cells_and_kzg_proofs_batch([]) == ([], [], [])
    assert ctx.blob_to_kzg_commitment_batch([]) == ([], [])
    assert ctx.recover_cells_and_kzg_proofs_batch([]) == ([], [], [])
    assert ctx.verify_cell_kzg_proof_batch([], [], [], []) is True
    assert ctx.verify_blob_kzg_proof_batch([], [], []) is True
    zero = b"\x00" * 131072
    cells, proofs = ctx.compute_cells_and_kzg_proofs(zero)       # all-zero blob: identity commitment and proofs
    assert all(p == b"\xc0" + b"\x00" * 47 for p in proofs) and all(c == b"\x00" * 2048 for c in cells)
    assert ctx.blob_to_kzg_commitment(zero) == b"\xc0" + b"\x00" * 47
    assert ctx.verify_cell_kzg_proof_batch([b"\xc0" + b"\x00" * 47] * 128, list(range(128)), cells, proofs) is True


# ------------------------------------------------------------------ many verifications in one call
def _expect_many(case_output):
    """(verified, ok) a problem must come back with: the reference's Ok(true) / Ok(false) / Err."""
    return (False, False) if case_output is None else (bool(case_output), True)


def test_verify_many_reproduces_every_vector_in_one_call(ctx):
    """All 30 reference vectors of verify_cell_kzg_proof_batch as ONE eth_kzg_amd_verify_cell_kzg_proof_batch_many call: every
    problem comes back with its own verdict -- true, false, or an error status where the single form raises -- and the
    malformed problems poison nothing around them.  Twice in a different order, and interleaved with empty problems."""
    cases = sorted(vectors.load("verify_cell_kzg_proof_batch").items())
    problems = [(c["input"]["commitments"], c["input"]["cell_indices"], c["input"]["cells"], c["input"]["proofs"]) for _, c in cases]
    for order in (list(range(len(cases))), list(reversed(range(len(cases))))):
        probs = []
        for k in order:
            probs.append(problems[k])
            probs.append(([], [], [], []))
        ver, st = ctx.verify_cell_kzg_proof_batch_many(probs)
        for j, k in enumerate(order):
            want_ver, want_ok = _expect_many(cases[k][1]["output"])
            assert (st[2 * j] == 0) == want_ok and ver[2 * j] == want_ver, (cases[k][0], st[2 * j], ver[2 * j])
            assert st[2 * j + 1] == 0 and ver[2 * j + 1] is True  # the empty problem verifies (verifier.rs:90-93)
    assert ctx.verify_cell_kzg_proof_batch_many([]) == ([], [])


def test_verify_many_matches_single_calls_and_oracle(ctx, oracle):
    """Synthetic problems of every shape in one call: one blob's 128 cells against one commitment (the reference's bench shape,
    benchmark-mt.rs:77-101), several blobs mixed, ragged subsets with duplicates, a single cell, tampered proofs / cells /
    commitments, identity points.  Verdicts equal the single-call form and the CPU oracle."""
    blobs = [synth.seeded_blob(70 + i) for i in range(4)] + [b"\x00" * 131072]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)

    def whole(b):
        return ([comms[b]] * 128, list(range(128)), list(cells[b]), list(proofs[b]))
    probs = [whole(b) for b in range(5)]
    C_, I_, L_, P_ = [], [], [], []
    for b in (0, 1, 2):
        for k in range(0, 128, 3):
            C_.append(comms[b]); I_.append(k); L_.append(cells[b][k]); P_.append(proofs[b][k])
    probs.append((C_, I_, L_, P_))                                               # three blobs interleaved
    sel = [3, 40, 41, 3, 127, 64, 64]
    probs.append(([comms[1]] * len(sel), sel, [cells[1][k] for k in sel], [proofs[1][k] for k in sel]))  # duplicates, any order
    probs.append(([comms[3]], [77], [cells[3][77]], [proofs[3][77]]))             # a single cell
    t = whole(0); t[3][5] = proofs[1][7]; probs.append(t)                         # tampered proof        -> false
    t = whole(1); t[2][100] = cells[0][1]; probs.append(t)                        # tampered cell         -> false
    t = whole(2); t[0][0] = comms[3]; probs.append(t)                             # swapped commitment    -> false
    t = whole(3); t[3][9] = b"\xc0" + bytes(47); probs.append(t)                  # identity as a proof   -> false
    t = whole(0); t[1][4] = 128; probs.append(t)                                  # index out of range    -> error
    t = whole(0); t[2][4] = b"\xff" * 2048; probs.append(t)                       # non-canonical element -> error
    t = whole(0); t[3][4] = b"\x80" + bytes(46) + b"\x05"; probs.append(t)        # a proof that is not a curve point / not in G1 -> error
    ver, stt = ctx.verify_cell_kzg_proof_batch_many(probs)
    for j, p in enumerate(probs):
        single = _call(ctx.verify_cell_kzg_proof_batch, *p)
        ref = _call(oracle.verify_cell_kzg_proof_batch, *p)
        assert single == ref, j
        want_ver, want_ok = _expect_many(ref)
        assert (stt[j] == 0) == want_ok and ver[j] == want_ver, (j, stt[j], ver[j], ref)
    assert ver[:8] == [True] * 8 and ver[8:12] == [False] * 4 and stt[12] == 3 and stt[13] == 1 and stt[14] == 2


def test_verify_many_large_call_in_chunks_and_concurrent_with_other_calls(ctx):
    """700 problems of 128 cells (three passes of at most 32768 cells) with a sprinkling of tampered ones, while another thread
    keeps proving and verifying on the same context: the path owns its lock, stream and scratch."""
    import threading
    blobs = [synth.seeded_blob(90 + i) for i in range(7)]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    probs, want = [], []
    for j in range(700):
        b = j % 7
        P = list(proofs[b])
        bad = j % 50 == 17
        if bad:
            P[j % 128] = proofs[(b + 1) % 7][j % 128]
        probs.append(([comms[b]] * 128, list(range(128)), list(cells[b]), P))
        want.append(not bad)
    run = ctx.prepare_verify_cell_kzg_proof_batch_many(probs)
    # every problem valid: ONE folded pairing check per pass decides them all (verify_many.hip); then the mixed call, whose
    # folded check fails and falls back to per-problem checks
    good = [p for p, w in zip(probs, want) if w]
    ver, stt = ctx.verify_cell_kzg_proof_batch_many(good)
    assert stt == [0] * len(good) and ver == [True] * len(good)
    stop, errors = threading.Event(), []

    def other():
        try:
            while not stop.is_set():
                assert ctx.compute_cells_and_kzg_proofs(blobs[0]) == (cells[0], proofs[0])
                assert ctx.verify_cell_kzg_proof_batch(*probs[0]) is True
        except Exception as e:  # pragma: no cover
            errors.append(e)
    th = threading.Thread(target=other)
    th.start()
    try:
        for _ in range(2):
            ver, stt = run()
            assert stt == [0] * 700 and ver == want
    finally:
        stop.set()
        th.join()
    assert not errors, errors


def test_verify_many_without_folding_gives_the_same_verdicts(ctx, monkeypatch):
    """ETH_KZG_AMD_VM_FOLD=0: one pairing check per problem, always (round 3's first form): same verdicts as the folded form on a
    mix of valid, invalid and malformed problems."""
    blobs = [synth.seeded_blob(140 + i) for i in range(3)]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    probs = []
    for j in range(40):
        b = j % 3
        P, L, I = list(proofs[b]), list(cells[b]), list(range(128))
        if j % 7 == 3:
            P[j] = proofs[(b + 1) % 3][j]
        if j == 20:
            I[0] = 200
        if j == 30:
            L[3] = b"\xff" * 2048
        probs.append(([comms[b]] * 128, I, L, P))
    folded = ctx.verify_cell_kzg_proof_batch_many(probs)
    monkeypatch.setenv("ETH_KZG_AMD_VM_FOLD", "0")
    plain = ctx.verify_cell_kzg_proof_batch_many(probs)
    assert folded == plain
    assert plain[1][20] == 3 and plain[1][30] == 1 and [v for j, v in enumerate(plain[0]) if j % 7 == 3] == [False] * 6
    monkeypatch.delenv("ETH_KZG_AMD_VM_FOLD")
    all_good = [p for j, p in enumerate(probs) if j % 7 != 3 and j not in (20, 30)]
    assert ctx.verify_cell_kzg_proof_batch_many(all_good) == ([True] * len(all_good), [0] * len(all_good))


def test_verify_many_finds_wrong_proofs_by_searching_folded_ranges(ctx, monkeypatch):
    """A many-verification call that contains wrong proofs: the folded pairing check of the pass fails, and the wrong problems are
    found by folding sub-ranges of the resident weighted sums (k_vm_fold_ranges) with one pairing per probe -- not by 1024
    pairings.  Verdicts must be exact per problem for one wrong proof (first, last, middle), neighbours, a cluster, every third
    problem (the search gives up and checks one by one), wrong proofs next to malformed problems -- and equal to what the
    per-problem re-check of round 3 says (ETH_KZG_AMD_VM_SEARCH=0 on a second context)."""
    blobs = [synth.seeded_blob(150 + i) for i in range(3)]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    n_prob, n_cells = 300, 16

    def problem(j, wrong=False, malformed=None):
        b = j % 3
        ks = [(j + 5 * t) % 128 for t in range(n_cells)]
        P = [proofs[b][k] for k in ks]
        L = [cells[b][k] for k in ks]
        I = list(ks)
        if wrong:
            P[j % n_cells] = proofs[(b + 1) % 3][ks[j % n_cells]]
        if malformed == "index":
            I[0] = 128
        if malformed == "cell":
            L[1] = b"\xff" * 2048
        return ([comms[b]] * n_cells, I, L, P)
    monkeypatch.setenv("ETH_KZG_AMD_VM_SEARCH", "0")
    plain_ctx = kzg.DASContext(use_precomp=True)
    monkeypatch.delenv("ETH_KZG_AMD_VM_SEARCH")
    try:
        for wrong in ([0], [n_prob - 1], [137], [40, 41], [7, 8, 9, 10, 11, 200], list(range(0, n_prob, 3)), list(range(n_prob)), []):
            ws = set(wrong)
            probs = [problem(j, wrong=j in ws, malformed="index" if j == 20 else "cell" if j == 260 else None) for j in range(n_prob)]
            got = ctx.verify_cell_kzg_proof_batch_many(probs)
            want_ver = [False if j in (20, 260) else j not in ws for j in range(n_prob)]
            want_st = [3 if j == 20 else 1 if j == 260 else 0 for j in range(n_prob)]
            assert got == (want_ver, want_st), (wrong[:8], [j for j in range(n_prob) if got[0][j] != want_ver[j]][:8])
            assert plain_ctx.verify_cell_kzg_proof_batch_many(probs) == got
    finally:
        plain_ctx.close()


def test_verify_many_search_level_larger_than_its_range_buffers(ctx):
    """8,200 one-cell problems with 1,150 scattered wrong proofs in ONE pass: the per-problem fallback has not triggered yet
    (suspects * 4 < live problems), K is clamped to 2 and a level needs more probes than the range buffers hold (2,048).  The
    level must run in batches -- round 4 threw "too many probes", the whole call returned ERR_DEVICE and every VALID problem
    lost its verdict (reachable by a caller fed adversarial proofs)."""
    blobs = [synth.seeded_blob(170 + i) for i in range(2)]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    n_prob = 8200
    wrong = set(range(3, n_prob, 7)) - set(range(0, n_prob, 64))
    wrong = set(sorted(wrong)[:1150])
    probs = []
    for j in range(n_prob):
        b, k = j & 1, (j * 37) % 128
        p = proofs[1 - b][k] if j in wrong else proofs[b][k]
        probs.append(([comms[b]], [k], [cells[b][k]], [p]))
    ver, status = ctx.verify_cell_kzg_proof_batch_many(probs)
    assert status == [0] * n_prob
    assert ver == [j not in wrong for j in range(n_prob)], [j for j in range(n_prob) if ver[j] != (j not in wrong)][:10]


def test_serial_paths_overlap_across_threads(ctx):
    """The verification / recovery / commitment entry points run on engine lanes created on demand (c_eth_kzg.h, "Threading"):
    four threads calling at once all get correct answers (the overlap itself is measured by bench.py, not asserted here)."""
    import threading
    blobs = [synth.seeded_blob(120 + i) for i in range(4)]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    runs = [ctx.prepare_verify_cell_kzg_proof_batch([comms[b]] * 128, list(range(128)), cells[b], proofs[b]) for b in range(4)]
    for r in runs:
        assert r() is True  # warm-up: creates nothing yet (one caller), loads code
    reps = 30
    errors = []

    def worker(b):
        try:
            for _ in range(reps):
                assert runs[b]() is True
                assert ctx.blob_to_kzg_commitment(blobs[b]) == comms[b]
                half = list(range(64))
                rc, rp = ctx.recover_cells_and_kzg_proofs(half, cells[b][:64])
                assert rc == cells[b] and rp == proofs[b]
        except Exception as e:  # pragma: no cover
            errors.append((b, e))
    ths = [threading.Thread(target=worker, args=(b,)) for b in range(4)]
    for _ in range(1):  # first round creates the lanes
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    assert not errors, errors

    def verify_only(b):
        try:
            for _ in range(reps):
                assert runs[b]() is True
        except Exception as e:  # pragma: no cover
            errors.append((b, e))
    # Correctness only: how much the lanes overlap is a measurement, and measurements live in bench.py
    # (configs.verify_128_cells_from_4_threads) -- a shared host's clock must not be able to turn the parity suite red.
    ths = [threading.Thread(target=verify_only, args=(b,)) for b in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors


def test_concurrent_single_verifications_are_combined(ctx):
    """eth_kzg_verify_cell_kzg_proof_batch from many threads at once, as the reference's Node binding drives one context
    (bindings/node/src/lib.rs:92-299): up to one caller per engine lane takes the latency path, callers beyond that are
    combined into many-verification passes (Engine::verify_cell_kzg_proof_batch_combined).  Every caller gets ITS verdict -- true, false or
    the error the single path raises -- whatever else rode in the same pass."""
    import threading
    import time
    blobs = [synth.seeded_blob(160 + i) for i in range(4)]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    kinds = []
    for b in range(4):
        good = ([comms[b]] * 128, list(range(128)), list(cells[b]), list(proofs[b]))
        badp = (good[0], good[1], good[2], [proofs[b][1]] + list(proofs[b][1:]))
        sub = ([comms[b]] * 5, [3, 9, 9, 100, 127], [cells[b][k] for k in (3, 9, 9, 100, 127)], [proofs[b][k] for k in (3, 9, 9, 100, 127)])
        malformed = (good[0], [128] + list(range(1, 128)), good[2], good[3])
        noncanon = (good[0], good[1], [b"\xff" * 2048] + list(cells[b][1:]), good[3])
        kinds += [(good, True), (badp, False), (sub, True), (malformed, None), (noncanon, None), (([], [], [], []), True)]
    prepared = [(ctx.prepare_verify_cell_kzg_proof_batch(*args), want) for args, want in kinds]
    errors, n_threads, reps = [], 12, 12

    def worker(t):
        try:
            for r in range(reps):
                run, want = prepared[(5 * t + 7 * r) % len(prepared)]
                try:
                    got = run()
                except kzg.KzgError:
                    got = None
                assert got == want, (t, r, got, want)
        except Exception as e:  # pragma: no cover
            errors.append(e)
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    assert not errors, errors[:3]
    print(f"{n_threads} threads x {reps} single verifications (mixed shapes): {n_threads * reps / dt:.0f} calls/s")


@pytest.mark.gpu
def test_one_lane_per_point_forms_with_the_quad_kernels_switched_off():
    """Verifications of a few hundred points run with four lanes per point (g1_coop.hpp); the one-lane forms they replaced stay in
    the library for large inputs and behind ETH_KZG_AMD_COOP_POINTS=0, which is read once per process: tests/coop_off_check.py
    runs the single path, a small many-verification pass and the EIP-4844 verifier on them, valid, tampered and with a
    proof outside the subgroup."""
    import subprocess
    import sys
    env = dict(os.environ, ETH_KZG_AMD_COOP_POINTS="0", ETH_KZG_AMD_TABLE_GB="8")  # small tables: the parent test process holds the wide ones
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "coop_off_check.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "coop-off ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
