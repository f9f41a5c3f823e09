"""Window tables: every GLV table width against the oracle at stage and
end-to-end level, the memory budget knob, and the progressive start -- a context serves from small start tables at once
and switches to the wide ones when a helper thread has built them; bytes never depend on the table in use.
Reference knob: UsePrecomp::Yes{width} (crates/cryptography/bls12_381/src/fixed_base_msm.rs:41-49); reference bench:
"Initialize context" (crates/eip7594/benches/benchmark-mt.rs:103-113).

Since round 5 the commitment table (the monomial SRS as 64 groups of 64 bases) is a GLV table of the same kind (half the groups of
the FK20 table: half its size at every width), so every width is also checked through blob_to_kzg_commitment at every batch regime.

This module holds no module-wide context: a 71 GB table does not fit next to the 242 GB the other tests' contexts hold."""
import ctypes as C
import importlib
import os
import sys
import threading
import time

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib
import synth
import test_gpu_fullsize as full

pytestmark = pytest.mark.gpu
kzg = importlib.import_module("rust-eth-kzg_amd")
INF = b"\xc0" + bytes(47)
GLV_ADDS = {16: 16, 15: 18, 14: 20, 12: 22, 8: 32}  # gathered additions per (scalar, base) = 2 ceil(128 / w)
GLV_GB = {16: 206.2, 15: 70.9, 14: 29.0, 12: 14.5, 8: 1.6}  # mixed window widths (launch.hpp): W windows cover the 128-bit half exactly
SRS_GB = {15: 35.4, 14: 14.5, 12: 7.3, 8: 0.8}            # the commitment table: 64 groups instead of 128 (never eight windows: 103 GB)


def _commitments_check(ctx, oracle, seed, sizes=(1, 5, 70, 600)):
    """blob_to_kzg_commitment over the context's commitment table in every MSM regime (one block per MSM, a lane per window,
    four chunks per MSM) against the oracle on a sample; blob 2 is the zero polynomial (the point at infinity)."""
    for n in sizes:
        blobs = full._random_blobs(n, seed + n)
        if n > 2:
            blobs[2] = 0
        st, comms = ctx.blob_to_kzg_commitment_batch([blobs[b].tobytes() for b in range(n)])
        assert list(st) == [0] * n
        if n > 2:
            assert comms[2] == INF
        for b in sorted({0, n // 2, n - 1}):
            assert comms[b] == oracle.blob_to_kzg_commitment(blobs[b].tobytes()), (n, b)


def _windows(w):
    """(lowest bit, width) of the W = ceil(128 / w) windows of a GLV table of nominal width w (launch.hpp: b = 128 // W bits each, the
    lowest 128 - W b windows one bit more)."""
    W = -(-128 // w)
    b, r = 128 // W, 128 - W * (128 // W)
    out, lo = [], 0
    for k in range(W):
        bits = b + (1 if k < r else 0)
        out.append((lo, bits))
        lo += bits
    assert lo == 128
    return out


def _msm_stage_check(ctx, tag):
    """128 fixed-base MSM_64 per scalar set through the stage hook against oracle_g1_msm (edge scalars included)."""
    lib = kzg.load_library()
    n_msm = 9
    sc = [synth.seeded_scalars(128 * 64, b"tab%d" % m) for m in range(n_msm)]
    w = ctx.window_bits()
    lam = 0xac45a4010001a40200000000ffffffff
    edge = [0, 1, synth.R - 1, 1 << (w - 1), (1 << (w - 1)) + 1, (1 << w) - 1, (1 << 254), (1 << (w * 3)) - (1 << (w - 1)),
            lam, lam - 1, lam + 1, (lam - 1) // 2, (lam + 1) // 2, (lam + 1) // 2 + 1, lam * ((lam + 1) // 2), synth.R - lam,
            (1 << 127), (1 << 128) - 1, (1 << 126) + (1 << (w - 1)), ((1 << 127) - 1) * lam % synth.R]
    # the windows of a GLV table have MIXED widths covering the 128-bit half exactly (launch.hpp: b = 128 // W bits, the lowest
    # 128 - W b windows one bit more): values on both sides of every window boundary, and digits at the ends of every window's table
    W = -(-128 // w)
    b, r = 128 // W, 128 - W * (128 // W)
    lo = 0
    for k in range(W):
        bits = b + (1 if k < r else 0)
        edge += [(1 << lo), (1 << lo) - 1, (1 << (lo + bits - 1)), (1 << (lo + bits - 1)) + (1 << lo), ((1 << (lo + bits - 1)) - 1) << 0,
                 ((1 << bits) - 1) << lo, (((1 << (lo + bits - 1)) | (1 << lo)) * lam) % synth.R]
        lo += bits
    assert lo == 128
    # HALF-scalars chosen directly: k = k1 + k2 lambda with |k1|, |k2| inside the balanced split's cell
    # (glv.hpp: magnitudes <= (lambda + 1) / 2 + 1 ~ 2^126.4, so the split returns exactly these halves), whose signed window digits
    # sit at the recoding's edges in EVERY window at once: the top bit of every window set (each digit becomes negative and carries
    # into the next window, through every 15 -> 14-bit boundary of a mixed-width table), the largest digits that do not carry, the
    # smallest that do, and all ones (a carry chain from bit 0 to the top window)
    cap = (1 << 126) - 1
    tops = sum(1 << (l + bt - 1) for l, bt in _windows(w)) & cap
    below = sum(((1 << (bt - 1)) - 1) << l for l, bt in _windows(w)) & cap
    above = sum(((1 << (bt - 1)) + 1) << l for l, bt in _windows(w)) & cap
    for h in (tops, below, above, cap):
        assert h < (lam - 1) // 2
        edge += [h, h * lam, h + h * lam, -h + h * lam, h - h * lam]
    assert len(edge) <= 64 * 3
    for k, v in enumerate(edge):
        sc[0][k] = (v % synth.R).to_bytes(32, "big")
    sc[1] = [bytes(32)] * (128 * 64)
    flat = b"".join(b"".join(s) for s in sc)
    out = C.create_string_buffer(n_msm * 128 * 48)
    assert lib.eth_kzg_amd_test_fixed_msm(ctx.handle, flat, n_msm, out) == 0
    got = lambda m, j: out.raw[(m * 128 + j) * 48:(m * 128 + j + 1) * 48]
    assert all(got(1, j) == INF for j in range(128)), tag
    for (m, j) in [(0, 0), (0, 1), (0, 2), (2, 64), (8, 77)]:
        pts = b"".join(full._fk20_base_column(i)[j] for i in range(64))
        scal = b"".join(sc[m][j * 64:(j + 1) * 64])
        assert got(m, j) == oracle_lib.g1_msm(pts, scal), (tag, m, j)


@pytest.mark.parametrize("width", [15, 14, 12, 8])
def test_every_glv_width_matches_oracle(oracle, monkeypatch, width):
    """ETH_KZG_AMD_GLV_WINDOW: the GLV table of that width (W = ceil(128 / w) windows per half), every MSM schedule at
    stage level and every batch regime end to end.  (Width 16 is the default table of every other GPU test.)"""
    full._torch_first()
    monkeypatch.setenv("ETH_KZG_AMD_GLV_WINDOW", str(width))
    keep = kzg.DASContext(use_precomp=True)  # holds the table while the schedule variants come and go
    try:
        assert keep.window_bits() == width and keep.tables_ready() == 1
        assert abs(keep.table_bytes() / 1e9 - GLV_GB[width] - SRS_GB[15]) < 0.02 * GLV_GB[width] + 0.5  # + the nine-window commitment table (the suite's budget is "max")
        for chunks in ("auto", "0", "4"):
            if chunks == "auto":
                monkeypatch.delenv("ETH_KZG_AMD_MSM_CHUNKS", raising=False)
            else:
                monkeypatch.setenv("ETH_KZG_AMD_MSM_CHUNKS", chunks)
            c2 = kzg.DASContext(use_precomp=True)
            try:
                assert c2.window_bits() == width
                _msm_stage_check(c2, (width, chunks))
            finally:
                c2.close()
        monkeypatch.delenv("ETH_KZG_AMD_MSM_CHUNKS", raising=False)
        # end to end: flat (<= 8 blobs), windowed, four chunks per MSM
        for n, seed in ((1, 1), (5, 2), (70, 3), (600, 4)):
            blobs = full._random_blobs(n, 7000 + 10 * width + seed)
            if n > 2:
                blobs[2] = 0
            st, cells, proofs = full._compute_on_device(keep, blobs)
            assert st == [0] * n
            full._check_sample_against_oracle(oracle, blobs, cells, proofs, sorted({0, min(2, n - 1), n // 2, n - 1}))
    finally:
        keep.close()


def test_table_budget_picks_the_widest_glv_table_that_fits(oracle, monkeypatch):
    """ETH_KZG_AMD_TABLE_GB bounds both tables together: the commitment table takes the widest GLV table within a third of it, the
    FK20 table the widest within the rest (a plain table of the same speed needs more than twice the memory).  Every commitment
    table width gives the oracle's commitments in every batch regime."""
    full._torch_first()
    for budget, want_w, want_srs in ((90, 15, 14), (80, 14, 14), (25, 12, 12), (3, 8, 8)):
        monkeypatch.setenv("ETH_KZG_AMD_TABLE_GB", str(budget))
        c = kzg.DASContext(use_precomp=True)
        try:
            assert c.window_bits() == want_w, (budget, c.window_bits())
            assert c.table_bytes() <= budget * 1e9
            assert abs(c.table_bytes() / 1e9 - GLV_GB[want_w] - SRS_GB[want_srs]) < 0.5, (budget, c.table_bytes())
            _commitments_check(c, oracle, 7600 + budget)
            blobs = full._random_blobs(20, 7700 + budget)
            st, cells, proofs = full._compute_on_device(c, blobs)
            assert st == [0] * 20
            full._check_sample_against_oracle(oracle, blobs, cells, proofs, [0, 19])
            assert c.blob_to_kzg_commitment(blobs[3].tobytes()) == oracle.blob_to_kzg_commitment(blobs[3].tobytes())
        finally:
            c.close()


def test_default_budget_is_a_stated_108_gb(oracle, monkeypatch):
    """Without ETH_KZG_AMD_TABLE_GB the window tables take at most 108 GB (the nine-window GLV tables: 70.9 GB for FK20, 35.4 GB for commitments), not
    whatever the HBM holds; "max" and a negative budget argument give the widest tables; the budget argument of
    eth_kzg_amd_das_context_try_new wins over the environment.  Same bytes on every table."""
    full._torch_first()
    blobs = full._random_blobs(20, 7811)
    monkeypatch.delenv("ETH_KZG_AMD_TABLE_GB", raising=False)
    for kwargs, want_w in (({}, 15), ({"table_budget_gb": 60}, 14), ({"table_budget_gb": -1}, 16)):
        c = kzg.DASContext(use_precomp=True, **kwargs)
        try:
            assert c.window_bits() == want_w, (kwargs, c.window_bits())
            if want_w == 15:
                assert 105e9 < c.table_bytes() <= 108e9, c.table_bytes()
                assert c.window_count() == 9
                _msm_stage_check(c, "default budget")  # the mixed 15 / 14-bit windows at stage level, edge half-scalars included
                _commitments_check(c, oracle, 7820, sizes=(3, 600))
            st, cells, proofs = full._compute_on_device(c, blobs)
            assert st == [0] * 20
            full._check_sample_against_oracle(oracle, blobs, cells, proofs, [0, 19])
            assert c.blob_to_kzg_commitment(blobs[3].tobytes()) == oracle.blob_to_kzg_commitment(blobs[3].tobytes())
        finally:
            c.close()


def test_context_next_to_190_gb_held_by_someone_else_comes_up_narrower(oracle, monkeypatch):
    """A host application that shares the GPU: with 190 GB of the 288 taken (here by a torch tensor of this process; the library only
    sees free memory) a context created with the default budget (108 GB) does not fail or abort -- it comes up on the widest tables
    that still fit (narrower than width 15) and gives the oracle's bytes."""
    import torch
    full._torch_first()
    monkeypatch.delenv("ETH_KZG_AMD_TABLE_GB", raising=False)
    for _ in range(100):  # the tables of the previous test's contexts are freed by their builder threads: wait until the memory is back
        if torch.cuda.mem_get_info()[0] > 270 * 10**9:
            break
        time.sleep(0.1)
    hog = torch.empty(190 * 10**9, dtype=torch.uint8, device="cuda")
    try:
        c = kzg.DASContext(use_precomp=True, table_budget_gb=0)  # through eth_kzg_amd_das_context_try_new: an error would be raised, not an abort
        try:
            assert c.window_bits() < 15 and c.table_bytes() < 90e9, (c.window_bits(), c.table_bytes())
            blobs = full._random_blobs(70, 7812)
            st, cells, proofs = full._compute_on_device(c, blobs)
            assert st == [0] * 70
            full._check_sample_against_oracle(oracle, blobs, cells, proofs, [0, 63, 64, 69])
        finally:
            c.close()
    finally:
        del hog
        torch.cuda.empty_cache()


def test_progressive_start_serves_at_once_and_switches_tables(oracle, monkeypatch):
    """eth_kzg_das_context_new returns on the start tables (GLV width 8 for FK20 and for commitments).  The wide tables are allocated in pieces
    of under a gigabyte and taken into use GROUP BY GROUP while the helper thread builds them (an MSM stage = the ready groups on
    the wide table + the rest on the start table): calls made before, during -- at every mix of the two tables the loop happens
    to meet -- and after the switch give identical bytes (and match the oracle), and a commitment / a verification issued from
    other threads during the build come back right.  How long a call waits during the build is a measurement: bench.py records it
    (configs.context_creation_s.calls_during_build); nothing here compares clocks."""
    import torch
    full._torch_first()
    monkeypatch.delenv("ETH_KZG_AMD_PROGRESSIVE", raising=False)
    t0 = time.perf_counter()
    c = kzg.DASContext(use_precomp=True, wait_tables=False)
    t_new = time.perf_counter() - t0
    try:
        first_state, first_w = c.tables_ready(0), c.window_bits()
        blobs = full._random_blobs(40, 4242)
        blobs[1] = 0
        st, cells0, proofs0 = full._compute_on_device(c, blobs)
        t_first = time.perf_counter() - t0
        assert st == [0] * 40
        comm0 = c.blob_to_kzg_commitment(blobs[5].tobytes())
        full._check_sample_against_oracle(oracle, blobs, cells0, proofs0, [0, 1, 39])
        assert comm0 == oracle.blob_to_kzg_commitment(blobs[5].tobytes())
        # other threads use the serial paths (commitment on the SRS table that is being widened too, verification) meanwhile
        side_errors, stop = [], threading.Event()
        cl = [cells0[5][k * 2048:(k + 1) * 2048].tobytes() for k in range(128)]
        pl = [proofs0[5][k * 48:(k + 1) * 48].tobytes() for k in range(128)]

        def side():
            try:
                while not stop.is_set():
                    assert c.blob_to_kzg_commitment(blobs[5].tobytes()) == comm0
                    assert c.verify_cell_kzg_proof_batch([comm0] * 128, list(range(128)), cl, pl) is True
            except Exception as e:  # pragma: no cover
                side_errors.append(e)
        th = threading.Thread(target=side)
        th.start()
        # keep calling (one call every ~5 ms) while the helper thread builds and publishes the wide tables
        widths_seen, groups_seen, n_calls, waits = {first_w}, [c.table_groups_ready()], 0, []
        try:
            while c.tables_ready(0) == 0 and n_calls < 20000:
                t1 = time.perf_counter()
                st, cells, proofs = full._compute_on_device(c, blobs)
                waits.append(time.perf_counter() - t1)
                assert st == [0] * 40 and np.array_equal(cells, cells0) and np.array_equal(proofs, proofs0), (n_calls, groups_seen[-1])
                widths_seen.add(c.window_bits())
                g = c.table_groups_ready()
                if c.tables_ready(0) == 0 and c.window_bits() == first_w:
                    assert g >= groups_seen[-1] or groups_seen[-1] == 128, (g, groups_seen[-5:])  # groups only ever become ready
                groups_seen.append(g)
                n_calls += 1
                time.sleep(0.005)
        finally:
            stop.set()
            th.join()
        assert not side_errors, side_errors
        assert c.tables_ready(-1) == 1
        assert c.window_bits() == 16 and c.table_groups_ready() == 128
        st, cells, proofs = full._compute_on_device(c, blobs)
        assert np.array_equal(cells, cells0) and np.array_equal(proofs, proofs0)
        assert c.blob_to_kzg_commitment(blobs[5].tobytes()) == comm0
        mixed = sorted({g for g in groups_seen if 0 < g < 128})
        print(f"progressive start: context_new {t_new:.2f} s, first 40-blob result after {t_first:.2f} s, start state {first_state} "
              f"on width {first_w}, {n_calls} calls during the build (longest {max(waits) * 1e3 if waits else 0:.0f} ms, median "
              f"{sorted(waits)[len(waits) // 2] * 1e3 if waits else 0:.1f} ms), widths seen {sorted(widths_seen)}, "
              f"{len(mixed)} distinct mixes of start and wide table met (groups ready: {mixed[:6]} ... {mixed[-3:]})")
        if first_state == 0:
            assert first_w == 8
            assert n_calls == 0 or mixed, "the wide table was never used before it was complete"
    finally:
        c.close()


def test_contexts_created_and_freed_while_the_tables_are_built(oracle, monkeypatch):
    """The window tables are shared by the contexts of a process and built by whichever comes first.  (a) A second context created
    while the first one's helper thread is still filling the wide tables uses their ready groups too, and ends on the same tables
    without a second 206 GB.  (b) A context freed in the middle of its build abandons it within one piece; what it had built is
    given back, and the next context builds the tables from scratch.  Bytes always equal to the oracle's."""
    import torch
    full._torch_first()
    monkeypatch.delenv("ETH_KZG_AMD_PROGRESSIVE", raising=False)
    blobs = full._random_blobs(24, 5151)
    blobs[3] = 0
    # (b) first: freed mid-build
    a = kzg.DASContext(use_precomp=True, wait_tables=False)
    st, cells0, proofs0 = full._compute_on_device(a, blobs)
    assert st == [0] * 24
    full._check_sample_against_oracle(oracle, blobs, cells0, proofs0, [0, 3, 23])
    for _ in range(200):  # let the build get somewhere (some groups ready), then free the context under it
        if a.table_groups_ready() >= 8 or a.tables_ready(0) != 0:
            break
        time.sleep(0.01)
    a.close()
    free_after_close = torch.cuda.mem_get_info()[0]
    assert free_after_close > 200e9, f"only {free_after_close / 1e9:.0f} GB free after the context was freed mid-build"
    # (a) two contexts, the second created while the first one builds
    a = kzg.DASContext(use_precomp=True, wait_tables=False)
    b = kzg.DASContext(use_precomp=True, wait_tables=False)
    try:
        seen_b = set()
        while a.tables_ready(0) == 0 or b.tables_ready(0) == 0:
            for c in (a, b):
                st, cells, proofs = full._compute_on_device(c, blobs)
                assert st == [0] * 24 and np.array_equal(cells, cells0) and np.array_equal(proofs, proofs0)
            seen_b.add(b.table_groups_ready())
            time.sleep(0.005)
        assert a.tables_ready(-1) == 1 and b.tables_ready(-1) == 1
        assert a.window_bits() == 16 and b.window_bits() == 16 and a.table_bytes() == b.table_bytes()
        assert torch.cuda.mem_get_info()[0] > 20e9, "the second context allocated tables of its own"
        for c in (a, b):
            st, cells, proofs = full._compute_on_device(c, blobs)
            assert np.array_equal(cells, cells0) and np.array_equal(proofs, proofs0)
        assert any(0 < g < 128 for g in seen_b), "the second context never used the first one's growing table"
        print(f"second context during the build: groups of the shared table it saw in use: {sorted(seen_b)[:5]} ... {sorted(seen_b)[-3:]}")
    finally:
        b.close()
        a.close()


def test_non_progressive_start_returns_on_the_final_tables(monkeypatch):
    full._torch_first()
    monkeypatch.setenv("ETH_KZG_AMD_PROGRESSIVE", "0")
    c = kzg.DASContext(use_precomp=True, wait_tables=False)
    try:
        assert c.tables_ready(0) == 1 and c.window_bits() == 16
    finally:
        c.close()
    c = kzg.DASContext(use_precomp=False, wait_tables=False)
    try:
        assert c.tables_ready(0) == 1 and c.window_bits() == 8 and c.table_bytes() < 2.5e9
    finally:
        c.close()
