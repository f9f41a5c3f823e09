import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
kzg = importlib.import_module("rust-eth-kzg_amd")
rng = np.random.RandomState(3)
blobs = rng.randint(0, 256, size=(2, 4096, 32), dtype=np.uint8)
blobs[:, :, 0] &= 0x3F
blobs = [blobs[i].tobytes() for i in range(2)]
ctx = kzg.DASContext(True)
st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
run = ctx.prepare_verify_cell_kzg_proof_batch([comms[0]] * 128, list(range(128)), cells[0], proofs[0])
for _ in range(10):
    assert run() is True
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    assert run() is True
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print("COOP_POINTS=%s: median %.3f ms, p10 %.3f, p90 %.3f" % (os.environ.get("ETH_KZG_AMD_COOP_POINTS"), ts[100], ts[20], ts[180]))
ctx.close()
