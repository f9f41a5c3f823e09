"""GPU parity at BASELINE.json's FULL sizes (the oracle cannot run these whole in seconds, so: every blob is held to the
size-independent invariants, a seeded sample of blobs is compared byte-for-byte with the CPU oracle).

  config 3  verify_cell_kzg_proof_batch, 64 blobs x 128 cells = 8192 cells   (crates/eip7594/tests/verify_cell_kzg_proof_batch.rs)
  config 4  compute_cells_and_kzg_proofs, 512-blob batch                      (crates/eip7594/tests/compute_cells_and_kzg_proofs.rs)
  config 5  recover_cells_and_kzg_proofs, 256 blobs at 50 % erasure           (crates/eip7594/tests/recover_cells_and_kzg_proofs.rs)
  bench     compute_cells_and_kzg_proofs, 2048 device-resident blobs (bench.py's step)
  widths    FK20 window tables of width 8 and 12 + the stage-level fixed-base MSM against oracle_g1_msm
            (crates/cryptography/bls12_381/src/fixed_base_msm_window.rs:178-321 tests widths 2..14)

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import importlib
import os

import numpy as np
import pytest

import oracle_lib
import synth
from conftest import TABLE_BUDGETS, table_budget_ctx

pytestmark = pytest.mark.gpu
kzg = importlib.import_module("rust-eth-kzg_amd")

INF = b"\xc0" + bytes(47)


def _torch_first():
    """torch initialises its HIP state before the engine creates its streams (the other order has failed to find the GPU)."""
    import torch
    torch.cuda.init()


@pytest.fixture(scope="module", params=TABLE_BUDGETS, ids=lambda b: f"tables-{b}")
def ctx(request):
    _torch_first()
    # every full-size test that takes `ctx` runs on the library's DEFAULT tables (108 GB) and on the widest ones (conftest.py)
    yield from table_budget_ctx(request.param, lambda: kzg.DASContext(use_precomp=True))


def _random_blobs(n, seed):
    """n blobs of uniformly random canonical field elements (top two bits cleared: < 2^254 < r)."""
    rng = np.random.RandomState(seed)
    a = rng.randint(0, 256, size=(n, 4096, 32), dtype=np.uint8)
    a[:, :, 0] &= 0x3F
    return a


def _compute_on_device(ctx, blobs_np):
    """blobs [n][4096][32] u8 -> (status, cells [n][128*2048] u8, proofs [n][128*48] u8) through the device entry point."""
    import torch
    n = blobs_np.shape[0]
    d_blobs = torch.from_numpy(blobs_np.reshape(-1)).cuda()
    d_cells = torch.empty(n * 128 * 2048, dtype=torch.uint8, device="cuda")
    d_proofs = torch.empty(n * 128 * 48, dtype=torch.uint8, device="cuda")
    st = ctx.compute_cells_and_kzg_proofs_device(n, d_blobs.data_ptr(), d_cells.data_ptr(), d_proofs.data_ptr())
    torch.cuda.synchronize()
    return st, d_cells.cpu().numpy().reshape(n, 128 * 2048), d_proofs.cpu().numpy().reshape(n, 128 * 48)


def _verify_every_proof(ctx, blobs_np, cells, proofs, tamper=None):
    """ALL n x 128 (commitment, cell, proof) triples of a batch through eth_kzg_amd_verify_cell_kzg_proof_batch_many, 1024
    problems (one blob's 128 cells each) per call -- what the reference does for all proofs at once
    (crates/cryptography/kzg_multi_open/src/fk20/prover.rs:312-378).  The verifier is pinned by the reference's 30 vectors, so
    this is a property check over the WHOLE batch at the size the bench times.  tamper = (blob, cell): that proof is replaced by
    its neighbour's (a valid G1 point, the wrong opening).  Returns the list of verdicts."""
    n = blobs_np.shape[0]
    st, commitments = ctx.blob_to_kzg_commitment_batch([blobs_np[b].tobytes() for b in range(n)])
    assert st == [0] * n
    verdicts = []
    idx = list(range(128))
    for lo in range(0, n, 1024):
        problems = []
        for b in range(lo, min(n, lo + 1024)):
            cb, pb = cells[b].tobytes(), proofs[b].tobytes()
            pl = [pb[48 * k:48 * k + 48] for k in range(128)]
            if tamper is not None and tamper[0] == b:
                pl[tamper[1]] = pl[(tamper[1] + 1) % 128]
            problems.append(([commitments[b]] * 128, idx, [cb[2048 * k:2048 * k + 2048] for k in range(128)], pl))
        ok, status = ctx.verify_cell_kzg_proof_batch_many(problems)
        assert status == [0] * len(problems), "a produced cell or proof failed to decode"
        verdicts += ok
    return verdicts


def _check_sample_against_oracle(oracle, blobs_np, cells, proofs, sample):
    for b in sample:
        ec, ep = oracle.compute_cells_and_kzg_proofs(blobs_np[b].tobytes())
        assert cells[b].tobytes() == b"".join(ec), f"blob {b}: cells differ from the oracle"
        assert proofs[b].tobytes() == b"".join(ep), f"blob {b}: proofs differ from the oracle"


def test_use_precomp_false_matches_oracle(oracle):
    """use_precomp = false (UsePrecomp::No, fixed_base_msm.rs:41-49): the 2.4 GB sixteen-window GLV tables and nothing wider -- the
    tables a use_precomp = true context starts on: identical bytes in the windowed and the chunked MSM schedule, commitments
    included.  (Rounds 1-4 had plain width-4 tables and their own kernels here.)"""
    width = 8
    _torch_first()
    c2 = kzg.DASContext(use_precomp=False)
    try:
        assert c2.tables_ready(0) == 1 and c2.window_bits() == width and 2.3e9 < c2.table_bytes() < 2.5e9
        blobs = _random_blobs(70, 800 + width)
        blobs[1] = 0
        st, cells, proofs = _compute_on_device(c2, blobs)
        assert st == [0] * 70
        _check_sample_against_oracle(oracle, blobs, cells, proofs, [0, 1, 63, 64, 69])
        # large-batch MSM schedule (threads own chunks of windows) at this width
        blobs = _random_blobs(1024, 900 + width)
        st, cells, proofs = _compute_on_device(c2, blobs)
        assert st == [0] * 1024
        _check_sample_against_oracle(oracle, blobs, cells, proofs, [0, 511, 1023])
        st, comms = c2.blob_to_kzg_commitment_batch([blobs[b].tobytes() for b in range(300)])
        assert list(st) == [0] * 300
        for b in (0, 150, 299):
            assert comms[b] == oracle.blob_to_kzg_commitment(blobs[b].tobytes())
    finally:
        c2.close()


@pytest.mark.parametrize("n", [321, 512, 2048])
def test_compute_full_batches_device_resident(ctx, oracle, n):
    """The bench step (2048 blobs), config 4's whole batch (512) and a chip-filling batch that is no multiple of the 64-lane
    groups (321: the last block of every MSM group is one lane wide) through the device entry point: data-in-first-half
    invariant (prover.rs:251-275) on EVERY blob, 16 sampled blobs (first, last, both sides of every 64-lane group
    boundary that the sample hits, the planted edge cases) byte-for-byte against the oracle, and -- at 512 and 2048 blobs --
    EVERY one of the n x 128 proofs verified against its commitment and cell by the many-verification entry point."""
    blobs = _random_blobs(n, 7000 + n)
    blobs[3] = 0                                                        # zero polynomial: identity proofs
    blobs[n - 2] = np.frombuffer(synth.dummy_blob(), dtype=np.uint8).reshape(4096, 32)
    blobs[65] = np.frombuffer((b"\x00" * 31 + b"\x07") * 4096, dtype=np.uint8).reshape(4096, 32)  # constant polynomial
    st, cells, proofs = _compute_on_device(ctx, blobs)
    assert st == [0] * n
    flat_in = blobs.reshape(n, 131072)
    assert np.array_equal(cells[:, :131072], flat_in), "cells 0..63 must be the blob itself"
    assert proofs[3].tobytes() == INF * 128 and proofs[65].tobytes() == INF * 128
    rng = np.random.RandomState(n)
    sample = {0, 3, 63, 64, 65, n // 2 - 1, n // 2, n - 2, n - 1}
    while len(sample) < 16:
        sample.add(int(rng.randint(n)))
    sample = sorted(sample)
    _check_sample_against_oracle(oracle, blobs, cells, proofs, sample)
    if n in (512, 2048):
        # EVERY proof of the batch, not the sample: n x 128 openings verified in passes of 1024 problems; then one proof of a
        # blob outside the sample (a lane group the oracle comparison did not touch) is replaced and exactly that problem fails
        assert _verify_every_proof(ctx, blobs, cells, proofs) == [True] * n
        touched = {x // 64 for x in sample}
        rest = [b for b in range(n // 3, n) if b not in sample]
        victim = next((b for b in rest if b // 64 not in touched), rest[0])  # a lane group the oracle sample did not touch, where there is one
        verdicts = _verify_every_proof(ctx, blobs, cells, proofs, tamper=(victim, 77))
        assert verdicts == [b != victim for b in range(n)], [b for b in range(n) if not verdicts[b]]
    # identical blobs in different lanes give identical results (lane independence across the whole batch)
    blobs2 = blobs.copy()
    blobs2[n - 1] = blobs2[0]
    st2, cells2, proofs2 = _compute_on_device(ctx, blobs2)
    assert st2 == [0] * n and np.array_equal(cells2[n - 1], cells[0]) and np.array_equal(proofs2[n - 1], proofs[0])
    assert np.array_equal(proofs2[: n - 1], proofs[: n - 1])


def test_six_thousand_blobs_in_one_device_resident_call(ctx, oracle):
    """a device-resident batch larger than one work set is meant to hold -- 6,000 blobs next to the window tables
    (242 GB of them under tables-max) -- runs as sub-batches on one stream (4096 + 1904; halved again should an allocation fail)
    instead of failing on hipMalloc: every blob held to the data-in-the-first-half invariant, a sample on both sides of the cut
    against the oracle, and every proof of the 1024 blobs around the cut verified."""
    n = 6000
    blobs = _random_blobs(n, 670)
    st, cells, proofs = _compute_on_device(ctx, blobs)
    assert st == [0] * n
    assert np.array_equal(cells[:, :131072], blobs.reshape(n, 131072)), "cells 0..63 must be the blob itself"
    _check_sample_against_oracle(oracle, blobs, cells, proofs, [0, 4095, 4096, 5999])
    assert _verify_every_proof(ctx, blobs[3584:4608], cells[3584:4608], proofs[3584:4608]) == [True] * 1024


def test_signed_and_unsigned_arena_give_identical_bytes(ctx, oracle, monkeypatch):
    """Round 6: above the circulant form (> 2 blobs) the G1 linear map's arena holds signed 13 x 30-bit points -- MSM sums (one block
    per MSM with its quad fold, windowed, chunked), constant multiplications (four, two or one lane per blob), additions / pairs /
    halved doubling runs (four lanes or one) and the compression's input all in csrc/fp30.hpp, g1_coop30.hpp: no conversions;
    ETH_KZG_AMD_ARENA_SIGNED=0 keeps the 14 x 29-bit arena and kernels of rounds 2-5.  Same bytes from both, at sizes on both sides
    of every schedule threshold (flat / windowed / chunked MSM, quad / pair / lane multiplications, the fused-pair schedule of
    >= 1024 lanes), with degenerate blobs inside -- the zero polynomial, a constant, sparse polynomials (identity operands, equal and
    opposite points meet the exact slow paths of add, add_sub and their quad forms) -- and through recovery."""
    monkeypatch.setenv("ETH_KZG_AMD_ARENA_SIGNED", "0")
    old = kzg.DASContext(use_precomp=True)  # shares ctx's tables; everything of the G1 linear map in the 14-digit field, as in rounds 2-5
    try:
        # up to 64 blobs: one lane group -- the several-lanes-per-blob kernels of the two fields (g1_coop30.hpp against g1_coop.hpp)
        for n in (6, 9, 16, 17, 20, 33, 64, 65, 129, 300, 1100):
            blobs = _random_blobs(n, 9100 + n)
            blobs[1] = 0
            blobs[2] = np.frombuffer((b"\x00" * 31 + b"\x05") * 4096, dtype=np.uint8).reshape(4096, 32)
            blobs[3] = np.frombuffer(_blob_from_coefficients([0] * 4095 + [11]), dtype=np.uint8).reshape(4096, 32)
            blobs[n // 2] = np.frombuffer(_blob_from_coefficients([3] + [0] * 63 + [9] + [0] * 4031), dtype=np.uint8).reshape(4096, 32)
            blobs[n - 1] = np.frombuffer(synth.dummy_blob(), dtype=np.uint8).reshape(4096, 32)
            a, b = _compute_on_device(ctx, blobs), _compute_on_device(old, blobs)
            assert a[0] == b[0] == [0] * n
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), n
            _check_sample_against_oracle(oracle, blobs, a[1], a[2], [0, 1, 2, 3, n // 2, n - 1])
        # one and two blobs: the circulant form (k_g1circ.hip) on both point forms, the one-block MSM with its quad fold before it
        for n in (1, 2):
            for kind in ("random", "degenerate"):
                blobs = _random_blobs(n, 9300 + n)
                if kind == "degenerate":
                    blobs[0] = np.frombuffer(_blob_from_coefficients([0] * 64 + [1] + [0] * 4031), dtype=np.uint8).reshape(4096, 32)
                    blobs[n - 1] = 0 if n == 1 else np.frombuffer((b"\x00" * 31 + b"\x05") * 4096, dtype=np.uint8).reshape(4096, 32)
                a, b = _compute_on_device(ctx, blobs), _compute_on_device(old, blobs)
                assert a[0] == b[0] == [0] * n
                assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (n, kind)
                _check_sample_against_oracle(oracle, blobs, a[1], a[2], list(range(n)))
        # commitments (MSM on the commitment table, fold over the 64 groups, compression) and the EIP-4844 proof, which is the same path
        for n in (1, 3, 70):
            cb = [bytes(x.tobytes()) for x in _random_blobs(n, 9400 + n)]
            cb[0] = bytes(131072)
            cb[n - 1] = synth.dummy_blob()
            ca, co = ctx.blob_to_kzg_commitment_batch(cb), old.blob_to_kzg_commitment_batch(cb)
            assert ca == co and ca[0] == [0] * n, n
            assert ca[1][n - 1] == oracle.blob_to_kzg_commitment(cb[n - 1]) and (n == 1 or ca[1][0] == b"\xc0" + bytes(47))
        z = (12345).to_bytes(32, "big")
        pr = ctx.compute_kzg_proof(cb[1], z)  # (the EIP-4844 vectors pin it; here: the same bytes from both point forms, and it verifies)
        assert pr == old.compute_kzg_proof(cb[1], z) and ctx.verify_kzg_proof(ca[1][1], z, pr[1], pr[0]) is True
        # recovery of 80 half-erased blobs runs the same map from coefficients
        import torch
        n = 80
        blobs = _random_blobs(n, 9200)
        st, cells, proofs = _compute_on_device(ctx, blobs)
        ext = torch.from_numpy(np.ascontiguousarray(cells)).cuda().view(n, 128, 2048).clone()
        ext[:, 1::2, :] = 0xEE
        outs = []
        for c in (ctx, old):
            d_c = torch.zeros(n * 128 * 2048, dtype=torch.uint8, device="cuda")
            d_p = torch.zeros(n * 128 * 48, dtype=torch.uint8, device="cuda")
            st = c.recover_cells_and_kzg_proofs_device(n, ext.data_ptr(), [list(range(0, 128, 2))] * n, d_c.data_ptr(), d_p.data_ptr())
            torch.cuda.synchronize()
            assert st == [0] * n
            outs.append((d_c.cpu().numpy().reshape(n, -1), d_p.cpu().numpy().reshape(n, -1)))
        assert np.array_equal(outs[0][0], cells) and np.array_equal(outs[0][1], proofs)
        assert np.array_equal(outs[1][0], cells) and np.array_equal(outs[1][1], proofs)
    finally:
        old.close()


def test_compute_512_through_the_host_batch_abi(ctx, oracle):
    """Config 4's 512 blobs through the host-pointer batch entry point (what a C / Go / Java caller uses):
    same bytes as the device-resident form."""
    n = 512
    blobs = _random_blobs(n, 7000 + n)
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch([blobs[b].tobytes() for b in range(n)])
    assert st == [0] * n
    _, dcells, dproofs = _compute_on_device(ctx, blobs)
    for b in range(n):
        assert b"".join(cells[b]) == dcells[b].tobytes() and b"".join(proofs[b]) == dproofs[b].tobytes(), b


@pytest.mark.parametrize("n", [1024, 1100, 2048])
def test_host_batch_with_the_msm_stage_in_two_launches(ctx, oracle, n):
    """From 1024 blobs the host-pointer path launches the MSMs of the first 256 blobs while the rest is still on the link and
    the others afterwards (Engine::run_proofs_from_coeffs, PROOFS_HEAD / PROOFS_TAIL): same bytes as the device-resident
    form (one launch), an invalid blob on either side of the cut reported and the others unaffected, a sample against the oracle."""
    blobs = _random_blobs(n, 9100 + n)
    bufs = ctx.host_batch_buffers(n)
    flat = np.ascontiguousarray(blobs.reshape(n, 131072))
    st = ctx.compute_cells_and_kzg_proofs_batch_np(flat, bufs)
    assert st == [0] * n
    _, dcells, dproofs = _compute_on_device(ctx, blobs)
    assert np.array_equal(bufs["cells"].reshape(n, -1), dcells.reshape(n, -1))
    assert np.array_equal(bufs["proofs"].reshape(n, -1), dproofs.reshape(n, -1))
    _check_sample_against_oracle(oracle, blobs, dcells, dproofs, [0, 255, 256, n - 1])
    good = bufs["proofs"].copy()
    bad = flat.copy()
    bad[100, :32] = 0xFF   # not canonical: before the cut
    bad[700, 32:64] = 0xFF  # after it
    st = ctx.compute_cells_and_kzg_proofs_batch_np(bad, bufs)
    assert [i for i, v in enumerate(st) if v != 0] == [100, 700]
    keep = np.ones(n, dtype=bool)
    keep[[100, 700]] = False
    assert np.array_equal(bufs["proofs"][keep], good[keep])


def _config3_inputs(ctx, n_blobs=64):
    blobs = _random_blobs(n_blobs, 3003)
    st, cells, proofs = _compute_on_device(ctx, blobs)
    assert st == [0] * n_blobs
    st2, comms = ctx.blob_to_kzg_commitment_batch([blobs[b].tobytes() for b in range(n_blobs)])
    assert st2 == [0] * n_blobs
    C, I, L, P = [], [], [], []
    for b in range(n_blobs):
        cb, pb = cells[b].tobytes(), proofs[b].tobytes()
        for k in range(128):
            C.append(comms[b]); I.append(k); L.append(cb[2048 * k:2048 * (k + 1)]); P.append(pb[48 * k:48 * (k + 1)])
    return blobs, comms, C, I, L, P


def test_verify_config3_full_size(ctx, oracle, monkeypatch):
    """64 blobs x 128 cells in ONE call (8192 cells, 64 distinct commitments, 17.6 MB of input): true; one tampered
    proof, one tampered cell, one swapped commitment: false each (not an error); verdicts equal the oracle's;
    the sharded form (partial + combine) at world 8 agrees.  At this size the lincombs run over byte-shifted point copies
    (k_verify.hip: k_pip_shift); a context with that form switched off must produce the same two pairing inputs."""
    blobs, comms, C, I, L, P = _config3_inputs(ctx)
    assert len(L) == 8192
    assert ctx.verify_cell_kzg_proof_batch(C, I, L, P) is True
    monkeypatch.setenv("ETH_KZG_AMD_PIP_SHIFT_MIN", "1000000")
    windowed = kzg.DASContext()
    try:
        assert windowed.verify_cell_kzg_proof_batch_partial(C, I, L, P, 0, 8192) == \
            ctx.verify_cell_kzg_proof_batch_partial(C, I, L, P, 0, 8192)
        assert windowed.verify_cell_kzg_proof_batch(C, I, L, P) is True
    finally:
        windowed.close()
    monkeypatch.delenv("ETH_KZG_AMD_PIP_SHIFT_MIN")
    assert oracle.verify_cell_kzg_proof_batch(C, I, L, P) is True
    P2 = list(P); P2[4097] = P[4098]
    L2 = list(L); L2[8191] = L[0]
    C2 = list(C); C2[700] = comms[63]
    for (c_, l_, p_) in ((C, L, P2), (C, L2, P), (C2, L, P)):
        assert ctx.verify_cell_kzg_proof_batch(c_, I, l_, p_) is False
    assert oracle.verify_cell_kzg_proof_batch(C, I, L, P2) is False
    # sharded over 8 ranks' worth of cell ranges
    world = 8
    bounds = [(r * 8192 // world, (r + 1) * 8192 // world) for r in range(world)]
    parts = [ctx.verify_cell_kzg_proof_batch_partial(C, I, L, P, lo, hi) for lo, hi in bounds]
    assert ctx.verify_cell_kzg_proof_batch_combine(parts) is True
    parts_bad = [ctx.verify_cell_kzg_proof_batch_partial(C, I, L, P2, lo, hi) for lo, hi in bounds]
    assert ctx.verify_cell_kzg_proof_batch_combine(parts_bad) is False
    # a malformed proof (not in G1) anywhere in the 8192 is an error, not "false"
    P3 = list(P); P3[5000] = b"\x80" + bytes(46) + b"\x05"
    if oracle_lib.g1_validate(P3[5000], True) != 0:
        with pytest.raises(kzg.KzgError):
            ctx.verify_cell_kzg_proof_batch(C, I, L, P3)


@pytest.mark.parametrize("pattern", ["even", "first_half"])
def test_recover_config5_full_size(ctx, oracle, pattern):
    """256 blobs at 50 % erasure, host-batch and device-resident forms: recovered cells and proofs == what the prover
    produced for the whole blob; 8 sampled blobs against the oracle's recovery."""
    import torch
    n = 256
    blobs = _random_blobs(n, 5005)
    st, cells, proofs = _compute_on_device(ctx, blobs)
    assert st == [0] * n
    idx = list(range(0, 128, 2)) if pattern == "even" else list(range(64))
    # device-resident form: extended blobs in HBM with the missing cells overwritten by junk
    flat = cells.reshape(n, 128, 2048).copy()
    missing = [k for k in range(128) if k not in idx]
    flat[:, missing, :] = 0xFF
    d_in = torch.from_numpy(flat.reshape(-1)).cuda()
    d_cells = torch.empty(n * 128 * 2048, dtype=torch.uint8, device="cuda")
    d_proofs = torch.empty(n * 128 * 48, dtype=torch.uint8, device="cuda")
    status = ctx.recover_cells_and_kzg_proofs_device(n, d_in.data_ptr(), [idx] * n, d_cells.data_ptr(), d_proofs.data_ptr())
    torch.cuda.synchronize()
    assert status == [0] * n
    assert np.array_equal(d_cells.cpu().numpy().reshape(n, -1), cells)
    assert np.array_equal(d_proofs.cpu().numpy().reshape(n, -1), proofs)
    # every one of the 256 x 128 recovered proofs opens its recovered cell against the blob's commitment (the whole batch, not a sample)
    assert _verify_every_proof(ctx, blobs, d_cells.cpu().numpy().reshape(n, -1), d_proofs.cpu().numpy().reshape(n, -1)) == [True] * n
    # host-batch form
    cell_bytes = [cells[b].tobytes() for b in range(n)]
    batch = [(idx, [cell_bytes[b][2048 * k:2048 * (k + 1)] for k in idx]) for b in range(n)]
    st2, rc, rp = ctx.recover_cells_and_kzg_proofs_batch(batch)
    assert st2 == [0] * n
    for b in range(n):
        assert b"".join(rc[b]) == cell_bytes[b] and b"".join(rp[b]) == proofs[b].tobytes(), b
    for b in sorted(np.random.RandomState(5).choice(n, 8, replace=False).tolist()):
        oc, op = oracle.recover_cells_and_kzg_proofs(batch[b][0], batch[b][1])
        assert rc[b] == oc and rp[b] == op, b


_FK20_BASES = {}


def _fk20_base_column(i):
    """S^[.][i]: FFT_128 of SRS vector i (batch_toeplitz.rs:46-61), compressed, computed by the oracle (cached)."""
    if i not in _FK20_BASES:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        raw = open(os.path.join(root, "rust-eth-kzg_amd", "data", "trusted_setup_4096.bin"), "rb").read()
        srs = [raw[16 + 48 * k:16 + 48 * (k + 1)] for k in range(4096)]
        vec = [srs[4096 - 1 - 64 - (i + 64 * pos)] if pos < 63 else INF for pos in range(128)]
        out = oracle_lib.g1_fft(b"".join(vec), inverse=False)
        _FK20_BASES[i] = [out[48 * j:48 * (j + 1)] for j in range(128)]
    return _FK20_BASES[i]


@pytest.mark.parametrize("precomp", [True, False])
@pytest.mark.parametrize("chunks", ["auto", "0", "4"])
def test_fixed_base_msm_stage_matches_oracle(oracle, monkeypatch, chunks, precomp):
    """Stage D alone (128 fixed-base MSM_64 per scalar set over the FK20 window tables) against oracle_g1_msm, for
    every MSM schedule: the windowed kernel (0) and four chunks of windows per MSM (4).  Scalars include 0,
    1, r-1 and values whose Booth digits hit the table ends (2^(c-1) and its negative)."""
    import ctypes as C
    if chunks != "auto":
        monkeypatch.setenv("ETH_KZG_AMD_MSM_CHUNKS", chunks)
    c2 = kzg.DASContext(use_precomp=precomp)  # False: the sixteen-window tables only
    try:
        lib = kzg.load_library()
        n_msm = 9  # > FLAT_MSM_MAX_SLICES so that the batch kernels run, not the one-block-per-MSM form
        sc = [synth.seeded_scalars(128 * 64, b"msm%d" % m) for m in range(n_msm)]
        w = c2.window_bits()
        lam = 0xac45a4010001a40200000000ffffffff  # the GLV eigenvalue: scalars around its multiples sit on the split's branch points
        edge = [0, 1, synth.R - 1, 1 << (w - 1), (1 << (w - 1)) + 1, (1 << w) - 1, (1 << 254), (1 << (w * 3)) - (1 << (w - 1)),
                lam, lam - 1, lam + 1, (lam - 1) // 2, (lam + 1) // 2, (lam + 1) // 2 + 1, lam * ((lam + 1) // 2), lam * ((lam + 1) // 2 + 1),
                lam * ((lam + 1) // 2) + (lam - 1) // 2, lam * ((lam + 1) // 2) + (lam + 1) // 2, synth.R - lam, (1 << 127), (1 << 128) - 1]
        for k, v in enumerate(edge):
            sc[0][k] = (v % synth.R).to_bytes(32, "big")
        sc[1] = [bytes(32)] * (128 * 64)  # all-zero scalars: identity everywhere
        flat = b"".join(b"".join(s) for s in sc)
        out = C.create_string_buffer(n_msm * 128 * 48)
        assert lib.eth_kzg_amd_test_fixed_msm(c2.handle, flat, n_msm, out) == 0
        got = lambda m, j: out.raw[(m * 128 + j) * 48:(m * 128 + j + 1) * 48]
        assert all(got(1, j) == INF for j in range(128))
        for (m, j) in [(0, 0), (0, 1), (2, 64), (5, 127), (8, 77)]:
            pts = b"".join(_fk20_base_column(i)[j] for i in range(64))
            scal = b"".join(sc[m][j * 64:(j + 1) * 64])
            assert got(m, j) == oracle_lib.g1_msm(pts, scal), (chunks, m, j)
    finally:
        c2.close()


@pytest.mark.parametrize("program", [0, 1, 2, 3, 4, 5])
def test_every_compilation_of_the_linear_map_matches_oracle(oracle, monkeypatch, program):
    """ETH_KZG_AMD_SLP_PROGRAM forces ONE compilation of the FK20 proofs map at every batch size (the engine picks by the number
    of 64-blob lane groups: the operation-count optimum with 350 constant multiplications when the chip is full, depth-optimised
    ones with 372 / 456 / 606 / 712 multiplications and shallower cheap levels when it is not; g1_linmap.hpp: Strategy).
    Same bytes from each of them at one lane group, two, and a ragged count, incl. the all-zero blob (identity everywhere)."""
    monkeypatch.setenv("ETH_KZG_AMD_SLP_PROGRAM", str(program))
    _torch_first()
    c2 = kzg.DASContext(use_precomp=True)
    try:
        for n in (9, 64, 130):
            blobs = _random_blobs(n, 900 + n)
            blobs[2] = 0
            st, cells, proofs = _compute_on_device(c2, blobs)
            assert st == [0] * n
            _check_sample_against_oracle(oracle, blobs, cells, proofs, [0, 2, min(63, n - 1), n - 1])
    finally:
        c2.close()


def test_pipelined_async_recoveries_on_a_user_stream(ctx):
    """Two back-to-back asynchronous device-resident recoveries on ONE user stream, the first one's input still being
    produced on that stream when the call is made (a device-to-device copy queued just before): the header promises
    stream order.  Results must equal the synchronous ones.  Then an asynchronous prover call and an asynchronous
    commitment call interleaved with them on the same stream."""
    import torch
    n = 48
    blobs = _random_blobs(2 * n, 6006)
    st, cells, proofs = _compute_on_device(ctx, blobs)
    assert st == [0] * (2 * n)
    idx = list(range(1, 128, 2))
    flat = cells.reshape(2 * n, 128, 2048).copy()
    flat[:, [k for k in range(128) if k not in idx], :] = 0xEE
    src = torch.from_numpy(flat.reshape(-1)).cuda()
    stream = torch.cuda.Stream()
    d_in = [torch.zeros(n * 128 * 2048, dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_c = [torch.empty(n * 128 * 2048, dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_p = [torch.empty(n * 128 * 48, dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_cc = torch.empty(n * 128 * 2048, dtype=torch.uint8, device="cuda")
    d_pp = torch.empty(n * 128 * 48, dtype=torch.uint8, device="cuda")
    d_blobs = torch.from_numpy(blobs[:n].reshape(-1)).cuda()
    d_comm = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for h in range(2):
            d_in[h].copy_(src[h * n * 128 * 2048:(h + 1) * n * 128 * 2048], non_blocking=True)  # input produced on the stream
            status = ctx.recover_cells_and_kzg_proofs_device(n, d_in[h].data_ptr(), [idx] * n, d_c[h].data_ptr(), d_p[h].data_ptr(),
                                                             stream=stream.cuda_stream)
            assert status == [0] * n
            if h == 0:
                ctx.compute_cells_and_kzg_proofs_device(n, d_blobs.data_ptr(), d_cc.data_ptr(), d_pp.data_ptr(), want_status=False,
                                                        stream=stream.cuda_stream)
                ctx.blob_to_kzg_commitment_device(n, d_blobs.data_ptr(), d_comm.data_ptr(), want_status=False, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    for h in range(2):
        assert np.array_equal(d_c[h].cpu().numpy().reshape(n, -1), cells[h * n:(h + 1) * n]), h
        assert np.array_equal(d_p[h].cpu().numpy().reshape(n, -1), proofs[h * n:(h + 1) * n]), h
    assert np.array_equal(d_cc.cpu().numpy().reshape(n, -1), cells[:n]) and np.array_equal(d_pp.cpu().numpy().reshape(n, -1), proofs[:n])
    _, comms = ctx.blob_to_kzg_commitment_batch([blobs[b].tobytes() for b in range(n)])
    assert d_comm.cpu().numpy().tobytes() == b"".join(comms)


def test_library_rccl_all_gather_at_world_one(ctx):
    """The library's own RCCL communicator (eth_kzg_amd_comm_unique_id / _comm_init / _all_gather), as far as a one-GPU
    box can take it: world = 1, ncclAllGather of the proof slab on a user stream must reproduce the slab."""
    import torch
    uid = kzg.DASContext.comm_unique_id()
    assert len(uid) == 128
    c2 = kzg.DASContext(use_precomp=True)
    try:
        c2.comm_init(uid, 0, 1)
        with pytest.raises(kzg.KzgError):
            c2.comm_init(uid, 0, 1)  # one communicator per context
        n = 8
        blobs = _random_blobs(n, 77)
        st, cells, proofs = _compute_on_device(c2, blobs)
        d_p = torch.from_numpy(proofs.reshape(-1)).cuda()
        d_all = torch.zeros_like(d_p)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            c2.all_gather(d_p.data_ptr(), d_all.data_ptr(), d_p.numel(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(d_all, d_p)
        c2.comm_destroy()
        with pytest.raises(kzg.KzgError):
            c2.all_gather(d_p.data_ptr(), d_all.data_ptr(), d_p.numel())
    finally:
        c2.close()


def test_single_process_fan_out_over_contexts(ctx):
    """eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi with two contexts on the one GPU of the test box (they share
    the window tables): slices 0..k and k..n run on two host threads concurrently; results equal the plain batch call."""
    n = 150
    blobs = np.ascontiguousarray(_random_blobs(n, 88).reshape(n, 131072))
    blobs[7] = 0xFF  # invalid blob in the first slice
    c2 = kzg.DASContext(use_precomp=True)
    try:
        bufs = ctx.host_batch_buffers(n)
        st = kzg.DASContext.compute_cells_and_kzg_proofs_batch_multi([ctx, c2], blobs, bufs)
        ref = ctx.host_batch_buffers(n)
        st_ref = ctx.compute_cells_and_kzg_proofs_batch_np(blobs, ref)
        assert st == st_ref and st[7] != 0 and sum(1 for x in st if x) == 1
        keep = [b for b in range(n) if b != 7]
        assert np.array_equal(bufs["cells"][keep], ref["cells"][keep]) and np.array_equal(bufs["proofs"][keep], ref["proofs"][keep])
    finally:
        c2.close()


def _blob_from_coefficients(coeffs):
    """blob (evaluation form, bit-reversed order, as the ABI takes it) of the polynomial with the given 4096 coefficients."""
    data = b"".join(int(c % synth.R).to_bytes(32, "big") for c in coeffs)
    ev = oracle_lib.fr_ntt(data, inverse=False)
    brp = lambda v: int(format(v, "012b")[::-1], 2)
    return b"".join(ev[32 * brp(k):32 * brp(k) + 32] for k in range(4096))


def test_sparse_polynomials_hit_the_degenerate_point_operations(ctx, oracle):
    """Polynomials with one or two non-zero coefficients: most of the 8192 FK20 scalars are zero, whole MSMs are the
    identity, the additions and doublings of the compiled linear map meet identity operands, equal and opposite points
    (their exact slow paths), and the constant multiplications meet the identity.  Batches of 12 (compiled map, windowed
    MSM), and the same blobs alone (circulant path, flat MSM) must all give the oracle's bytes."""
    def poly(*terms):
        c = [0] * 4096
        for idx, val in terms:
            c[idx] = val
        return c
    polys = [poly((0, 1)), poly((63, 5)), poly((64, 1)), poly((65, synth.R - 1)), poly((4032, 7)), poly((4095, 1)),
             poly((64, 1), (128, 1)), poly((64, 1), (128, synth.R - 1)), poly((100, 3), (4000, 9)), [1] * 4096,
             poly(*[(64 * m, 1) for m in range(64)]), poly(*[(i, i + 1) for i in range(64)])]
    blobs = [_blob_from_coefficients(c) for c in polys]
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    assert st == [0] * len(blobs)
    for b, blob in enumerate(blobs):
        ec, ep = oracle.compute_cells_and_kzg_proofs(blob)
        assert cells[b] == ec and proofs[b] == ep, f"polynomial {b} in the batch"
        c1, p1 = ctx.compute_cells_and_kzg_proofs(blob)
        assert c1 == ec and p1 == ep, f"polynomial {b} alone"
    # and in a batch that fills lane groups unevenly (70 = 64 + 6 lanes) next to random blobs
    rnd = _random_blobs(58, 4242)
    mixed = blobs + [rnd[i].tobytes() for i in range(58)]
    st, cells2, proofs2 = ctx.compute_cells_and_kzg_proofs_batch(mixed)
    assert st == [0] * 70 and cells2[:12] == cells and proofs2[:12] == proofs


def test_degenerate_blobs_inside_a_batch_that_fills_the_chip(ctx, oracle):
    """The large-batch forms of the point kernels have their own handling of identity / equal / opposite operands: k_g1_compress<4>
    (four positions share one inversion from 128 x 2048 lanes on; an identity takes part as Z = 1) and the fused a + b / a - b pair
    of the linear map with its exact slow path (from 1024 lanes on).  Zero, constant and one- / two-term polynomials embedded at
    scattered places of a 2112-blob device-resident batch (33 lane groups, a ragged count for the four-position compressor) must
    give the oracle's bytes, and their random neighbours too."""
    def poly(*terms):
        c = [0] * 4096
        for idx, val in terms:
            c[idx] = val
        return c
    special = {0: [0] * 4096, 1: poly((0, 7)), 63: poly((0, 1), (5, 3)), 64: poly((64, 1), (128, synth.R - 1)), 1000: poly((4095, 1)),
               1023: [1] * 4096, 1024: poly((64, 1), (128, 1)), 2047: poly(*[(64 * m, 1) for m in range(64)]), 2048: poly((63, 5)), 2111: [0] * 4096}
    n = 2112
    blobs = _random_blobs(n, 77077)
    for b, c in special.items():
        blobs[b] = np.frombuffer(_blob_from_coefficients(c), dtype=np.uint8).reshape(4096, 32)
    st, cells, proofs = _compute_on_device(ctx, blobs)
    assert st == [0] * n
    _check_sample_against_oracle(oracle, blobs, cells, proofs, sorted(special) + [2, 62, 65, 1001, 2110])
    inf = np.frombuffer((b"\xc0" + bytes(47)) * 128, dtype=np.uint8)
    for b in (0, 1, 2111):  # zero and constant polynomials: every proof is the identity
        assert np.array_equal(proofs[b], inf)


def test_a_large_many_verification_call_split_over_the_pass_slots(ctx):
    """A call of 256 problems x 128 cells is cut into three parts that run as passes of their own on the three pass slots at once
    (verify_many.hip: staging and hashing of one part under the GPU work of another).  Wrong proofs on both sides of the cuts,
    a malformed problem and an empty one: verdicts and statuses per problem as if nothing had been cut."""
    nb = 8
    blobs = _random_blobs(nb, 8181)
    st, cells, proofs = _compute_on_device(ctx, blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch([blobs[b].tobytes() for b in range(nb)])
    cl = [[cells[b][k * 2048:(k + 1) * 2048].tobytes() for k in range(128)] for b in range(nb)]
    pl = [[proofs[b][k * 48:(k + 1) * 48].tobytes() for k in range(128)] for b in range(nb)]
    wrong = {0, 84, 85, 86, 169, 170, 171, 255}
    probs = []
    for j in range(256):
        b = j % nb
        P = list(pl[b])
        if j in wrong:
            P[j % 128] = pl[(b + 1) % nb][j % 128]
        I = list(range(128))
        if j == 100:
            I[5] = 128
        probs.append(([comms[b]] * 128, I, cl[b], P) if j != 200 else ([], [], [], []))
    ver, stt = ctx.verify_cell_kzg_proof_batch_many(probs)
    assert stt == [3 if j == 100 else 0 for j in range(256)]
    assert ver == [False if j == 100 else j not in wrong for j in range(256)]
    good = [p for j, p in enumerate(probs) if j not in wrong and j != 100]
    assert ctx.verify_cell_kzg_proof_batch_many(good) == ([True] * len(good), [0] * len(good))


def test_verify_device_resident_matches_the_host_form(ctx, oracle):
    """eth_kzg_amd_verify_cell_kzg_proof_batch_device on the prover's own device buffers (config 3 at full size: cells
    and proofs never leave HBM before the call): true; a flipped proof byte pair, a swapped cell, a wrong commitment:
    false; a malformed proof: error; an empty batch: true."""
    import torch
    nb = 64
    blobs = _random_blobs(nb, 3003)
    d_blobs = torch.from_numpy(blobs.reshape(-1)).cuda()
    d_cells = torch.empty(nb * 128 * 2048, dtype=torch.uint8, device="cuda")
    d_proofs = torch.empty(nb * 128 * 48, dtype=torch.uint8, device="cuda")
    d_comm = torch.empty(nb * 48, dtype=torch.uint8, device="cuda")
    assert ctx.compute_cells_and_kzg_proofs_device(nb, d_blobs.data_ptr(), d_cells.data_ptr(), d_proofs.data_ptr()) == [0] * nb
    assert ctx.blob_to_kzg_commitment_device(nb, d_blobs.data_ptr(), d_comm.data_ptr()) == [0] * nb
    n = nb * 128
    d_c = d_comm.view(nb, 1, 48).expand(nb, 128, 48).contiguous().view(-1)  # one commitment per cell
    d_i = torch.arange(128, dtype=torch.int64, device="cuda").repeat(nb)
    torch.cuda.synchronize()
    run = lambda c=d_c, i=d_i, l=d_cells, p=d_proofs: ctx.verify_cell_kzg_proof_batch_device(n, c.data_ptr(), i.data_ptr(), l.data_ptr(), p.data_ptr())
    assert run() is True
    p2 = d_proofs.clone(); p2[48 * 77:48 * 78] = d_proofs[48 * 78:48 * 79]
    assert run(p=p2) is False
    l2 = d_cells.clone(); l2[2048 * 5:2048 * 6] = d_cells[2048 * 6:2048 * 7]
    assert run(l=l2) is False
    c2 = d_c.clone(); c2[:48] = d_c[48 * 128:48 * 129]
    assert run(c=c2) is False
    p3 = d_proofs.clone(); p3[:48] = 0
    with pytest.raises(kzg.KzgError):
        run(p=p3)
    assert ctx.verify_cell_kzg_proof_batch_device(0, 0, 0, 0, 0) is True
    # the verdicts against the CPU oracle directly (not only through expected booleans), on a 3-blob sub-batch:
    # the same device buffers go to the device entry point, their bytes to the oracle
    m = 3 * 128

    def oracle_verdict(c, l, p):
        cb, lb, pb = c[:48 * m].cpu().numpy().tobytes(), l[:2048 * m].cpu().numpy().tobytes(), p[:48 * m].cpu().numpy().tobytes()
        args = ([cb[48 * k:48 * (k + 1)] for k in range(m)], [k % 128 for k in range(m)], [lb[2048 * k:2048 * (k + 1)] for k in range(m)],
                [pb[48 * k:48 * (k + 1)] for k in range(m)])
        try:
            return oracle.verify_cell_kzg_proof_batch(*args)
        except oracle_lib.OracleError:
            return None
    for c, l, p in ((d_c, d_cells, d_proofs), (d_c, d_cells, p2), (d_c, l2, d_proofs), (c2, d_cells, d_proofs), (d_c, d_cells, p3)):
        want = oracle_verdict(c, l, p)
        try:
            got = ctx.verify_cell_kzg_proof_batch_device(m, c.data_ptr(), d_i.data_ptr(), l.data_ptr(), p.data_ptr())
        except kzg.KzgError:
            got = None
        assert got == want
