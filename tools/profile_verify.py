import importlib, os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
kzg = importlib.import_module("rust-eth-kzg_amd")
ctx = kzg.DASContext(True)
nb = 64
rng = np.random.RandomState(7)
a = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8); a[:, :, 0] &= 0x3F
blobs = [a[i].tobytes() for i in range(nb)]
st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
C, I, L, P = [], [], [], []
for b in range(nb):
    for k in range(128):
        C.append(comms[b]); I.append(k); L.append(cells[b][k]); P.append(proofs[b][k])
for r in range(5):
    assert ctx.verify_cell_kzg_proof_batch(C, I, L, P)
