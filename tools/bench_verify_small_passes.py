"""eth_kzg_amd_verify_cell_kzg_proof_batch_many with 1 .. 64 problems of 128 cells per call: milliseconds per call (the passes that
concurrent single calls are combined into).  python tools/bench_verify_small_passes.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")
rng = np.random.RandomState(7)
blobs = rng.randint(0, 256, size=(8, 4096, 32), dtype=np.uint8)
blobs[:, :, 0] &= 0x3F
blobs = [blobs[i].tobytes() for i in range(8)]
ctx = kzg.DASContext(True, wait_tables=False)
st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
probs = [([comms[b]] * 128, list(range(128)), cells[b], proofs[b]) for b in range(8)]
for B in (1, 2, 4, 8, 16, 32, 64):
    run = ctx.prepare_verify_cell_kzg_proof_batch_many([probs[j % 8] for j in range(B)])
    ver, stt = run()
    assert ver == [True] * B and stt == [0] * B
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{B:3d} problems per call: {ts[len(ts) // 2] * 1e3:6.2f} ms  ({B / ts[len(ts) // 2]:7.0f} verifications/s)", flush=True)
ctx.close()
