import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
kzg = importlib.import_module("rust-eth-kzg_amd")
rng = np.random.RandomState(3)
blobs = rng.randint(0, 256, size=(2, 4096, 32), dtype=np.uint8)
blobs[:, :, 0] &= 0x3F
blobs = [blobs[i].tobytes() for i in range(2)]
ctx = kzg.DASContext(True)
st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
run = ctx.prepare_verify_cell_kzg_proof_batch([comms[0]] * 128, list(range(128)), cells[0], proofs[0])
for _ in range(5):
    assert run() is True
os.environ["ETH_KZG_AMD_TRACE"] = "1"
for _ in range(3):
    t0 = time.perf_counter()
    assert run() is True
    print("call: %.3f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
ctx.close()
