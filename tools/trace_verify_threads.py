"""N threads x reps single verifications (128 cells, one commitment) on one context; meant to run under
rocprofv3 --hip-trace --kernel-trace (tools/README.md): python3 tools/trace_verify_threads.py N reps"""
import importlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")
n_thr, reps = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(7)
blobs = rng.randint(0, 256, size=(4, 4096, 32), dtype=np.uint8)
blobs[:, :, 0] &= 0x3F
blobs = [blobs[i].tobytes() for i in range(4)]
ctx = kzg.DASContext(True, wait_tables=False)
st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
runs = [ctx.prepare_verify_cell_kzg_proof_batch([comms[b]] * 128, list(range(128)), cells[b], proofs[b]) for b in range(4)]
for r in runs:
    assert r() is True


def hammer(r):
    for _ in range(reps):
        assert r()


for rnd in range(2):  # the first round creates the lanes
    ths = [threading.Thread(target=hammer, args=(runs[t % 4],)) for t in range(n_thr)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    print(f"round {rnd}: {n_thr} threads, {n_thr * reps / (time.perf_counter() - t0):.0f} verifications/s", flush=True)
ctx.close()
