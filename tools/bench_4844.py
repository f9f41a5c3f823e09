"""Latency of the EIP-4844 single-point operations through the C ABI (Python mirror)."""
import importlib, os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
kzg = importlib.import_module("rust-eth-kzg_amd")
ctx = kzg.DASContext(True)
rng = np.random.RandomState(11)
nb = 64
a = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8); a[:, :, 0] &= 0x3F
blobs = [a[i].tobytes() for i in range(nb)]
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
def timed(f, reps=5):
    f(); t = time.time()
    for _ in range(reps): r = f()
    return (time.time() - t) / reps * 1e3, r
out = {}
z = (12345).to_bytes(32, "big")
out["compute_kzg_proof_ms"], (pf, y) = timed(lambda: ctx.compute_kzg_proof(blobs[0], z))
out["verify_kzg_proof_ms"], ok = timed(lambda: ctx.verify_kzg_proof(comms[0], z, y, pf)); assert ok
out["compute_blob_kzg_proof_ms"], bp = timed(lambda: ctx.compute_blob_kzg_proof(blobs[0], comms[0]))
out["verify_blob_kzg_proof_ms"], ok = timed(lambda: ctx.verify_blob_kzg_proof(blobs[0], comms[0], bp)); assert ok
proofs = [ctx.compute_blob_kzg_proof(blobs[i], comms[i]) for i in range(nb)]
out["verify_blob_kzg_proof_batch_64_ms"], ok = timed(lambda: ctx.verify_blob_kzg_proof_batch(blobs, comms, proofs)); assert ok
out["blob_to_kzg_commitment_ms"], _ = timed(lambda: ctx.blob_to_kzg_commitment(blobs[0]))
print(json.dumps({k: round(v, 3) for k, v in out.items()}))
