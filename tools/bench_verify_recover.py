"""Timing of BASELINE.json configs 3 (verify 64 blobs x 128 cells) and 5 (recover, 50 % erasure) through the C ABI."""
import importlib, os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()  # before the engine touches HIP
kzg = importlib.import_module("rust-eth-kzg_amd")
ctx = kzg.DASContext(True)
nb = int(os.environ.get("NB", "64"))
rng = np.random.RandomState(7)
a = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8); a[:, :, 0] &= 0x3F
blobs = [a[i].tobytes() for i in range(nb)]
t = time.time(); st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs); t_cp = time.time() - t
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
C, I, L, P = [], [], [], []
for b in range(nb):
    for k in range(128):
        C.append(comms[b]); I.append(k); L.append(cells[b][k]); P.append(proofs[b][k])
out = {"blobs": nb, "cells": len(L), "compute_batch_host_api_s": round(t_cp, 3)}
ts = []
for r in range(3):
    t = time.time(); ok = ctx.verify_cell_kzg_proof_batch(C, I, L, P); ts.append(time.time() - t); assert ok
out["verify_s"] = [round(x, 4) for x in ts]
P2 = list(P); P2[77] = P[78]
t = time.time(); ok = ctx.verify_cell_kzg_proof_batch(C, I, L, P2); out["verify_tampered_s"] = round(time.time() - t, 4); assert ok is False
ts = []
for b in range(min(nb, 8)):
    idx = list(range(0, 128, 2))
    t = time.time(); rc, rp = ctx.recover_cells_and_kzg_proofs(idx, [cells[b][i] for i in idx]); ts.append(time.time() - t)
    assert rc == cells[b] and rp == proofs[b]
out["recover_even_cells_per_blob_s"] = [round(x, 4) for x in ts]
print(json.dumps(out))
nbr = min(nb, 256)
batch = [(list(range(0, 128, 2)), [cells[b][i] for i in range(0, 128, 2)]) for b in range(nbr)]
t = time.time(); st, rc, rp = ctx.recover_cells_and_kzg_proofs_batch(batch); dt = time.time() - t
assert st == [0] * nbr and all(rc[b] == cells[b] and rp[b] == proofs[b] for b in range(nbr))
print(json.dumps({"recover_batch_blobs": nbr, "recover_batch_s": round(dt, 3), "blobs_per_s": round(nbr / dt, 1)}))

# sharded verification run on one GPU: per-slice partial times (what each of `world` ranks would spend) and the combine
sh = importlib.import_module("rust-eth-kzg_amd.sharding")
for world in (2, 8):
    tp = []
    parts = []
    for r in range(world):
        lo, hi = sh.shard_bounds(len(L), world, r)
        t = time.time(); parts.append(ctx.verify_cell_kzg_proof_batch_partial(C, I, L, P, lo, hi)); tp.append(time.time() - t)
    t = time.time(); ok = ctx.verify_cell_kzg_proof_batch_combine(parts); tc = time.time() - t
    assert ok
    print(json.dumps({"verify_sharded_world": world, "partial_s_max": round(max(tp), 4), "partial_s_min": round(min(tp), 4), "combine_s": round(tc, 4)}))

# compute_cells fast path: device-resident blobs -> cells only, no FK20 stages
B = int(os.environ.get("NB_CELLS", "2048"))
dev = torch.device("cuda", 0)
src = torch.from_numpy(np.frombuffer(b"".join(blobs), dtype=np.uint8).copy()).to(dev)
d_blobs = src.repeat((B + nb - 1) // nb)[: B * 131072].contiguous()
d_cells = torch.empty(B * 128 * 2048, dtype=torch.uint8, device=dev)
for _ in range(2):
    ctx.compute_cells_and_kzg_proofs_device(B, d_blobs.data_ptr(), d_cells.data_ptr(), 0, want_status=False)
torch.cuda.synchronize()
t = time.time()
reps = 10
for _ in range(reps):
    ctx.compute_cells_and_kzg_proofs_device(B, d_blobs.data_ptr(), d_cells.data_ptr(), 0, want_status=False)
torch.cuda.synchronize()
dt = (time.time() - t) / reps
assert bytes(d_cells[:128 * 2048].cpu().numpy()) == b"".join(cells[0])
gb = B * (131072 + 262144) / 1e9
print(json.dumps({"compute_cells_blobs": B, "ms": round(dt * 1e3, 3), "blobs_per_s": round(B / dt), "algorithmic_GB_per_s": round(gb / dt, 1)}))

# device-resident recovery (config 5 for pipelines that hold the extended blobs in HBM): 50 % erasure, even cells present
nbr = min(nb, 256)
flat = np.frombuffer(b"".join(b"".join(cells[b]) for b in range(nbr)), dtype=np.uint8).copy()
d_in = torch.from_numpy(flat).to(dev)
d_oc = torch.empty(nbr * 128 * 2048, dtype=torch.uint8, device=dev)
d_op = torch.empty(nbr * 128 * 48, dtype=torch.uint8, device=dev)
pat = [list(range(0, 128, 2))] * nbr
for _ in range(2):
    st = ctx.recover_cells_and_kzg_proofs_device(nbr, d_in.data_ptr(), pat, d_oc.data_ptr(), d_op.data_ptr())
torch.cuda.synchronize()
t = time.time()
for _ in range(5):
    st = ctx.recover_cells_and_kzg_proofs_device(nbr, d_in.data_ptr(), pat, d_oc.data_ptr(), d_op.data_ptr())
torch.cuda.synchronize()
dt = (time.time() - t) / 5
assert st == [0] * nbr and bytes(d_op[:128 * 48].cpu().numpy()) == b"".join(proofs[0]) and torch.equal(d_oc, d_in)
print(json.dumps({"recover_device_blobs": nbr, "ms": round(dt * 1e3, 2), "blobs_per_s": round(nbr / dt)}))
