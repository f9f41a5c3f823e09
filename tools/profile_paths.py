"""A loop of one non-prover path, to put under `rocprofv3 --kernel-trace --stats -- python3 tools/profile_paths.py <path>`:
  verify   BASELINE config 3: verify_cell_kzg_proof_batch, 64 blobs x 128 cells, 5 calls
  recover  BASELINE config 5: recover_cells_and_kzg_proofs, 256 blobs at 50 % erasure (even cells), device-resident form, 3 calls
  4844     the EIP-4844 single-point operations, 5 calls each
  single   the reference's own criterion shapes (benchmark-mt.rs:36-101): one blob through compute, verify (128 cells, one
           commitment) and worst-case recovery (first 64 cells), 7 calls each
  verify_many  1024 independent verifications of 128 cells (64 distinct problems repeated) in ONE call of
           eth_kzg_amd_verify_cell_kzg_proof_batch_many, 4 calls
Prints one JSON line with host-side timings of the same calls."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")


def main():
    import torch
    what = sys.argv[1] if len(sys.argv) > 1 else "verify"
    ctx = kzg.DASContext(True)
    nb = {"verify": 64, "recover": 256, "4844": 4, "single": 1, "verify_many": 64}[what]
    rng = np.random.RandomState(7)
    a = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8)
    a[:, :, 0] &= 0x3F
    d_blobs = torch.from_numpy(a.reshape(-1)).cuda()
    d_cells = torch.empty(nb * 128 * 2048, dtype=torch.uint8, device="cuda")
    d_proofs = torch.empty(nb * 128 * 48, dtype=torch.uint8, device="cuda")
    assert ctx.compute_cells_and_kzg_proofs_device(nb, d_blobs.data_ptr(), d_cells.data_ptr(), d_proofs.data_ptr()) == [0] * nb
    torch.cuda.synchronize()
    out = {"path": what}
    if what == "verify":
        cells = d_cells.cpu().numpy().tobytes()
        proofs = d_proofs.cpu().numpy().tobytes()
        _, comms = ctx.blob_to_kzg_commitment_batch([a[b].tobytes() for b in range(nb)])
        C, I, L, P = [], [], [], []
        for b in range(nb):
            for k in range(128):
                j = b * 128 + k
                C.append(comms[b]); I.append(k); L.append(cells[2048 * j:2048 * (j + 1)]); P.append(proofs[48 * j:48 * (j + 1)])
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            assert ctx.verify_cell_kzg_proof_batch(C, I, L, P)
            ts.append(time.perf_counter() - t)
        out.update(cells=len(L), verify_ms=[round(x * 1e3, 2) for x in ts])
    elif what == "verify_many":
        cells = d_cells.cpu().numpy().tobytes()
        proofs = d_proofs.cpu().numpy().tobytes()
        _, comms = ctx.blob_to_kzg_commitment_batch([a[b].tobytes() for b in range(nb)])
        probs = []
        for b in range(nb):
            probs.append(([comms[b]] * 128, list(range(128)), [cells[2048 * (b * 128 + k):2048 * (b * 128 + k + 1)] for k in range(128)],
                          [proofs[48 * (b * 128 + k):48 * (b * 128 + k + 1)] for k in range(128)]))
        run = ctx.prepare_verify_cell_kzg_proof_batch_many([probs[j % nb] for j in range(1024)])
        ts = []
        for _ in range(4):
            t = time.perf_counter()
            ver, st = run()
            ts.append(time.perf_counter() - t)
            assert all(ver) and not any(st)
        out.update(problems=1024, cells_per_problem=128, call_ms=[round(x * 1e3, 2) for x in ts], verifications_per_s=round(1024 / min(ts)))
    elif what == "recover":
        idx = list(range(0, 128, 2))
        flat = d_cells.view(nb, 128, 2048).clone()
        flat[:, 1::2, :] = 0xFF
        d_oc = torch.empty_like(d_cells)
        d_op = torch.empty_like(d_proofs)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            st = ctx.recover_cells_and_kzg_proofs_device(nb, flat.data_ptr(), [idx] * nb, d_oc.data_ptr(), d_op.data_ptr())
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
            assert st == [0] * nb
        assert torch.equal(d_oc, d_cells) and torch.equal(d_op, d_proofs)
        out.update(blobs=nb, recover_ms=[round(x * 1e3, 2) for x in ts])
    elif what == "single":
        blob = a[0].tobytes()
        comm = ctx.blob_to_kzg_commitment(blob)
        cells, proofs = ctx.compute_cells_and_kzg_proofs(blob)
        run_v = ctx.prepare_verify_cell_kzg_proof_batch([comm] * 128, list(range(128)), cells, proofs)
        t = {}
        for name, fn in (("compute_cells_and_kzg_proofs", lambda: ctx.compute_cells_and_kzg_proofs(blob)),
                         ("verify_128_cells_one_commitment", run_v),
                         ("recover_first_64_cells", lambda: ctx.recover_cells_and_kzg_proofs(list(range(64)), cells[:64]))):
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
            t[name] = round(min(ts) * 1e3, 3)
        assert run_v() and ctx.recover_cells_and_kzg_proofs(list(range(64)), cells[:64]) == (cells, proofs)
        out.update(t)
    else:
        blob = a[0].tobytes()
        z = (12345).to_bytes(32, "big")
        comm = ctx.blob_to_kzg_commitment(blob)
        t = {}
        for name, fn in (("compute_kzg_proof", lambda: ctx.compute_kzg_proof(blob, z)),
                         ("compute_blob_kzg_proof", lambda: ctx.compute_blob_kzg_proof(blob, comm))):
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
            t[name] = round(min(ts) * 1e3, 2)
        proof, y = ctx.compute_kzg_proof(blob, z)
        bproof = ctx.compute_blob_kzg_proof(blob, comm)
        for name, fn in (("verify_kzg_proof", lambda: ctx.verify_kzg_proof(comm, z, y, proof)),
                         ("verify_blob_kzg_proof", lambda: ctx.verify_blob_kzg_proof(blob, comm, bproof))):
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); assert fn(); ts.append(time.perf_counter() - t0)
            t[name] = round(min(ts) * 1e3, 2)
        out.update(t)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
