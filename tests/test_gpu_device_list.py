"""A context over a DEVICE LIST behind the unchanged ABI, the hardened batched entry points (item 6) and the
constructor that cleans up after itself.

The reference's hosts create ONE context and share it between their threads (bindings/c/src/lib.rs:79-92,
bindings/node/src/lib.rs:35,75): with ETH_KZG_AMD_DEVICES (or eth_kzg_amd_das_context_new_on_devices) that one pointer owns an engine
per listed GPU -- single calls go to the least-loaded device, host-pointer batches are cut into contiguous slices, device-resident
calls run where the caller's buffers are.  The test box has one GPU: the list is `0,0` -- two engines on one device, sharing its
window tables, exactly the code path of two GPUs minus the second piece of silicon.  Bytes must equal a one-device context's.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import ctypes as C
import importlib
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import synth
import test_gpu_fullsize as full
import vectors

pytestmark = pytest.mark.gpu
kzg = importlib.import_module("rust-eth-kzg_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pair():
    """(one-device context, context over the device list 0,0), both on the library's default tables (shared through the registry)."""
    full._torch_first()
    saved = os.environ.pop("ETH_KZG_AMD_TABLE_GB", None)
    one = multi = None
    try:
        one = kzg.DASContext(use_precomp=True)
        multi = kzg.DASContext(use_precomp=True, devices=[0, 0])
        yield one, multi
    finally:
        for c in (multi, one):
            if c is not None:
                c.close()
        if saved is not None:
            os.environ["ETH_KZG_AMD_TABLE_GB"] = saved


def test_device_list_context_reports_its_devices_and_shares_tables(pair):
    one, multi = pair
    assert one.devices() == [0] and multi.devices() == [0, 0]
    assert multi.window_bits() == one.window_bits() == 15 and multi.window_count() == 9
    assert multi.table_bytes() == one.table_bytes()  # two engines on one GPU: its tables counted once
    assert multi.tables_ready(0) == 1
    assert kzg.load_library().eth_kzg_amd_abi_version() == 6


def test_compute_batch_is_cut_into_slices_with_identical_bytes(pair, oracle):
    one, multi = pair
    n = 151  # odd: slices of 75 and 76 blobs
    blobs = np.ascontiguousarray(full._random_blobs(n, 611).reshape(n, 131072))
    blobs[7] = 0xFF    # an invalid blob in the first slice
    blobs[140] = 0xFF  # and one in the second
    got, ref = multi.host_batch_buffers(n), one.host_batch_buffers(n)
    st = multi.compute_cells_and_kzg_proofs_batch_np(blobs, got)
    st_ref = one.compute_cells_and_kzg_proofs_batch_np(blobs, ref)
    assert st == st_ref and [b for b in range(n) if st[b]] == [7, 140]
    keep = [b for b in range(n) if b not in (7, 140)]
    assert np.array_equal(got["cells"][keep], ref["cells"][keep]) and np.array_equal(got["proofs"][keep], ref["proofs"][keep])
    for b in (0, 75, 76, 150):  # both sides of the cut against the oracle
        ec, ep = oracle.compute_cells_and_kzg_proofs(blobs[b].tobytes())
        assert got["cells"][b].tobytes() == b"".join(ec) and got["proofs"][b].tobytes() == b"".join(ep)
    # fewer blobs than devices: the empty slice is skipped
    assert multi.compute_cells_and_kzg_proofs_batch([blobs[3].tobytes()]) == one.compute_cells_and_kzg_proofs_batch([blobs[3].tobytes()])
    assert multi.compute_cells_and_kzg_proofs_batch([]) == one.compute_cells_and_kzg_proofs_batch([])


def test_commitment_and_recovery_batches_fan_out(pair, oracle):
    one, multi = pair
    blobs = [full._random_blobs(1, 620 + i)[0].tobytes() for i in range(9)]
    assert multi.blob_to_kzg_commitment_batch(blobs) == one.blob_to_kzg_commitment_batch(blobs)
    assert multi.blob_to_kzg_commitment_batch(blobs)[1][8] == oracle.blob_to_kzg_commitment(blobs[8])
    # recovery: five blobs with different erasure patterns, one of them unrecoverable (63 cells)
    st, cells, proofs = one.compute_cells_and_kzg_proofs_batch(blobs[:5])
    patterns = [list(range(0, 128, 2)), list(range(64)), list(range(64, 128)), list(range(63)), list(range(128))]
    batch = [(idx, [cells[b][i] for i in idx]) for b, idx in enumerate(patterns)]
    got = multi.recover_cells_and_kzg_proofs_batch(batch)
    ref = one.recover_cells_and_kzg_proofs_batch(batch)
    assert got[0] == ref[0]
    st_r, cells_r, proofs_r = got
    for b in (0, 1, 2, 4):
        assert cells_r[b] == ref[1][b] and proofs_r[b] == ref[2][b]
    assert [bool(x) for x in st_r] == [False, False, False, True, False]
    for b in (0, 1, 2, 4):
        assert cells_r[b] == cells[b] and proofs_r[b] == proofs[b]


def test_verify_many_is_cut_by_problem(pair):
    one, multi = pair
    cases = sorted(vectors.load("verify_cell_kzg_proof_batch").items())
    problems = [(c["input"]["commitments"], c["input"]["cell_indices"], c["input"]["cells"], c["input"]["proofs"]) for _, c in cases]
    assert len(problems) == 30
    assert multi.verify_cell_kzg_proof_batch_many(problems) == one.verify_cell_kzg_proof_batch_many(problems)
    assert multi.verify_cell_kzg_proof_batch_many(problems[:1]) == one.verify_cell_kzg_proof_batch_many(problems[:1])


def test_single_calls_from_many_threads_spread_over_the_list(pair, oracle):
    """The reference's usage: one context, many threads, one problem per call.  Every result right; and the vectors through the
    sixteen unchanged symbols give what they give on one device."""
    one, multi = pair
    blobs = [synth.seeded_blob(700 + i) for i in range(8)]
    want = [one.compute_cells_and_kzg_proofs(b) for b in blobs]
    assert want[0] == tuple(oracle.compute_cells_and_kzg_proofs(blobs[0]))
    out, errs = [None] * 8, []

    def work(i):
        try:
            r = multi.compute_cells_and_kzg_proofs(blobs[i])
            assert multi.blob_to_kzg_commitment(blobs[i]) == one.blob_to_kzg_commitment(blobs[i])
            idx = list(range(0, 128, 2))
            rec = multi.recover_cells_and_kzg_proofs(idx, [r[0][k] for k in idx])
            assert rec == r
            assert multi.verify_cell_kzg_proof_batch([one.blob_to_kzg_commitment(blobs[i])] * 4, [0, 5, 64, 127],
                                                     [r[0][k] for k in (0, 5, 64, 127)], [r[1][k] for k in (0, 5, 64, 127)])
            out[i] = r
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert out == want
    for fam, call in (("compute_cells_and_kzg_proofs", lambda c, i: c.compute_cells_and_kzg_proofs(i["blob"])),
                      ("blob_to_kzg_commitment", lambda c, i: c.blob_to_kzg_commitment(i["blob"]))):
        for name, case in sorted(vectors.load(fam).items())[:6]:
            def run(c):
                try:
                    return call(c, case["input"])
                except kzg.KzgError:
                    return None
            assert run(multi) == run(one), (fam, name)


def test_device_resident_calls_run_where_the_buffers_are(pair):
    one, multi = pair
    blobs = full._random_blobs(70, 640)
    st_m, cells_m, proofs_m = full._compute_on_device(multi, blobs)
    st_o, cells_o, proofs_o = full._compute_on_device(one, blobs)
    assert st_m == st_o == [0] * 70 and np.array_equal(cells_m, cells_o) and np.array_equal(proofs_m, proofs_o)


def test_environment_device_list_through_the_unchanged_constructor(oracle):
    """ETH_KZG_AMD_DEVICES=0,0 + eth_kzg_das_context_new(bool) -- what a host written against the reference calls."""
    code = (
        "import importlib, os, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import torch; torch.cuda.init()\n"
        "import synth\n"
        "kzg = importlib.import_module('rust-eth-kzg_amd')\n"
        "c = kzg.DASContext(use_precomp=True)\n"
        "assert c.devices() == [0, 0], c.devices()\n"
        "blobs = [synth.seeded_blob(750 + i) for i in range(5)]\n"
        "st, cells, proofs = c.compute_cells_and_kzg_proofs_batch(blobs)\n"
        "import hashlib\n"
        "print('DIGEST', hashlib.sha256(b''.join(b''.join(x) for x in cells) + b''.join(b''.join(x) for x in proofs)).hexdigest())\n"
        "c.close()\n" % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, ETH_KZG_AMD_DEVICES="0,0", ETH_KZG_AMD_TABLE_GB="3")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "DIGEST" in out.stdout, out.stdout + out.stderr
    import hashlib
    want = hashlib.sha256()
    res = [oracle.compute_cells_and_kzg_proofs(synth.seeded_blob(750 + i)) for i in range(5)]
    want.update(b"".join(b"".join(r[0]) for r in res) + b"".join(b"".join(r[1]) for r in res))
    assert out.stdout.split("DIGEST")[1].split()[0] == want.hexdigest()


# ------------------------------------------------------------------------------------------------ hardening
def test_oversized_counts_with_a_live_context(pair):
    one, _ = pair
    lib = kzg.load_library()
    for n in (1 << 32, (1 << 32) + 5, (1 << 24) + 1):
        res = lib.eth_kzg_amd_compute_cells_and_kzg_proofs_batch(one.handle, n, None, None, None, None)
        assert res.status == 1 and C.string_at(res.error_msg).decode().startswith("InvalidInput")
        lib.eth_kzg_free_error_message(res.error_msg)
        res = lib.eth_kzg_amd_compute_cells_and_kzg_proofs_device(one.handle, n, None, None, None, None, None)
        assert res.status == 1
        lib.eth_kzg_free_error_message(res.error_msg)
    # the context is none the worse for it
    assert one.blob_to_kzg_commitment(synth.dummy_blob())


def test_device_batch_is_cut_into_sub_batches(oracle):
    """A device-resident call larger than ETH_KZG_AMD_DEVICE_BATCH_MAX blobs runs as sub-batches on one stream (the default bound is
    4096; here 128, so that 300 blobs are three sub-batches of 128 + 128 + 44): bytes equal the uncut call's."""
    full._torch_first()
    blobs = full._random_blobs(300, 660)
    blobs[129] = 0xFF
    saved = {k: os.environ.get(k) for k in ("ETH_KZG_AMD_DEVICE_BATCH_MAX", "ETH_KZG_AMD_TABLE_GB")}
    os.environ["ETH_KZG_AMD_TABLE_GB"] = "3"
    try:
        os.environ["ETH_KZG_AMD_DEVICE_BATCH_MAX"] = "128"
        cut = kzg.DASContext(use_precomp=True)
        os.environ.pop("ETH_KZG_AMD_DEVICE_BATCH_MAX")
        whole = kzg.DASContext(use_precomp=True)
        try:
            a, b = full._compute_on_device(cut, blobs), full._compute_on_device(whole, blobs)
            assert a[0] == b[0] and [i for i, s in enumerate(a[0]) if s] == [129]
            keep = [i for i in range(300) if i != 129]
            assert np.array_equal(a[1][keep], b[1][keep]) and np.array_equal(a[2][keep], b[2][keep])
            full._check_sample_against_oracle(oracle, blobs, a[1], a[2], [0, 127, 128, 255, 256, 299])
            # the commitment and the recovery device forms are cut the same way
            import torch
            d_blobs = torch.from_numpy(blobs.reshape(-1)).cuda()
            outs = []
            for c in (cut, whole):
                d_comm = torch.zeros(300 * 48, dtype=torch.uint8, device="cuda")
                st = c.blob_to_kzg_commitment_device(300, d_blobs.data_ptr(), d_comm.data_ptr())
                torch.cuda.synchronize()
                outs.append((st, d_comm.cpu().numpy().reshape(300, 48)[keep]))
            assert outs[0][0] == outs[1][0] and [i for i, s in enumerate(outs[0][0]) if s] == [129] and np.array_equal(outs[0][1], outs[1][1])
            assert outs[0][1][0].tobytes() == oracle.blob_to_kzg_commitment(blobs[0].tobytes())
            # recovery: the even cells of every blob (the odd ones overwritten with junk), blob 200 with too few cells
            ext = torch.from_numpy(np.ascontiguousarray(b[1])).cuda().view(300, 128, 2048).clone()
            ext[:, 1::2, :] = 0xFF
            present = [list(range(0, 128, 2))] * 300
            present[200] = list(range(0, 126, 2))
            rec = []
            for c in (cut, whole):
                d_c = torch.zeros(300 * 128 * 2048, dtype=torch.uint8, device="cuda")
                d_p = torch.zeros(300 * 128 * 48, dtype=torch.uint8, device="cuda")
                st = c.recover_cells_and_kzg_proofs_device(300, ext.data_ptr(), present, d_c.data_ptr(), d_p.data_ptr())
                torch.cuda.synchronize()
                rec.append((st, d_c.cpu().numpy().reshape(300, -1), d_p.cpu().numpy().reshape(300, -1)))
            good = [i for i in range(300) if i not in (129, 200)]
            assert rec[0][0] == rec[1][0] and rec[0][0][200] == 3 and all(rec[0][0][i] == 0 for i in good)
            assert np.array_equal(rec[0][1][good], b[1][good]) and np.array_equal(rec[0][2][good], b[2][good])
            assert np.array_equal(rec[1][1][good], b[1][good]) and np.array_equal(rec[1][2][good], b[2][good])
        finally:
            cut.close()
            whole.close()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# ------------------------------------------------------------------------------------------------ try_new never kills the host
def test_constructor_that_throws_late_cleans_up_and_the_next_context_works(oracle):
    """ETH_KZG_AMD_FAULT=constructor makes the engine's constructor throw after its last step but one -- streams, events, constants,
    the start tables and (before round 6) the table builder thread all exist by then.  try_new must return NULL + a message, the
    process must live (a joinable std::thread in a half-built object is std::terminate), the memory must come back, and the next
    context must work.  In a child process: a regression is an abort, which must fail this test and not the run."""
    code = (
        "import importlib, os, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import torch; torch.cuda.init()\n"
        "import synth\n"
        "kzg = importlib.import_module('rust-eth-kzg_amd')\n"
        "def failed(**kw):\n"
        "    os.environ['ETH_KZG_AMD_FAULT'] = 'constructor'\n"
        "    try:\n"
        "        kzg.DASContext(use_precomp=True, table_budget_gb=30, **kw)\n"
        "        raise SystemExit('the injected fault did not fire')\n"
        "    except kzg.KzgError as e:\n"
        "        assert 'injected fault' in str(e), str(e)\n"
        "    finally:\n"
        "        del os.environ['ETH_KZG_AMD_FAULT']\n"
        "def good():\n"
        "    kzg.DASContext(use_precomp=True, table_budget_gb=30, wait_tables=False).close()\n"
        "free = lambda: torch.cuda.mem_get_info()[0]\n"
        "failed(); good(); failed(devices=[0, 0])   # warm: what the HIP runtime itself keeps per process (code objects, queue scratch) is in\n"
        "m0 = free()\n"
        "for k in range(4): good()\n"
        "m1 = free()\n"
        "for k in range(3): failed()\n"
        "failed(devices=[0, 0])\n"
        "m2 = free()\n"
        "print('GROWTH four good cycles %%.2f GB, four failed constructors %%.2f GB' %% ((m0 - m1) / 1e9, (m1 - m2) / 1e9))\n"
        "assert (m1 - m2) <= max(m0 - m1, 0) + 0.3e9, 'failed constructors keep memory that completed ones give back'\n"

        "c = kzg.DASContext(use_precomp=True, table_budget_gb=30)\n"
        "assert c.window_bits() == 12, c.window_bits()  # 30 GB: eleven windows for both tables\n"
        "import hashlib\n"
        "r = c.compute_cells_and_kzg_proofs(synth.seeded_blob(780))\n"
        "print('DIGEST', hashlib.sha256(b''.join(r[0]) + b''.join(r[1])).hexdigest())\n"
        "c.close()\n" % (ROOT, os.path.join(ROOT, "tests")))
    env = {k: v for k, v in os.environ.items() if k != "ETH_KZG_AMD_TABLE_GB"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "DIGEST" in out.stdout, out.stdout + out.stderr
    print([l for l in out.stdout.splitlines() if l.startswith("GROWTH")])
    import hashlib
    r = oracle.compute_cells_and_kzg_proofs(synth.seeded_blob(780))
    assert out.stdout.split("DIGEST")[1].split()[0] == hashlib.sha256(b"".join(r[0]) + b"".join(r[1])).hexdigest()
