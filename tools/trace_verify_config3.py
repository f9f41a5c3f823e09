"""Config 3 (64 blobs x 128 cells in one eth_kzg_verify_cell_kzg_proof_batch call) with the library's own lap trace
(ETH_KZG_AMD_TRACE=1): how long the transcript hash takes and what is left behind it."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")
nb = 64
rng = np.random.RandomState(3)
blobs = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8)
blobs[:, :, 0] &= 0x3F
blobs = [blobs[i].tobytes() for i in range(nb)]
ctx = kzg.DASContext(True)
st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
C, I, L, P = [], [], [], []
for b in range(nb):
    for k in range(128):
        C.append(comms[b]); I.append(k); L.append(cells[b][k]); P.append(proofs[b][k])
run = ctx.prepare_verify_cell_kzg_proof_batch(C, I, L, P)
for _ in range(3):
    assert run() is True
os.environ["ETH_KZG_AMD_TRACE"] = "1"
for _ in range(3):
    t0 = time.perf_counter()
    assert run() is True
    print("call: %.3f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
ctx.close()
