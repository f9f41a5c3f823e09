"""Single eth_kzg_verify_cell_kzg_proof_batch calls (the reference bench's shape: 128 cells, one commitment) from N host threads
on ONE context -- the reference's usage model (bindings/node/src/lib.rs:92-299): verifications per second at N = 1, 2, 4, 8, 16, 32.
usage: python tools/bench_verify_threads.py [reps]   (env: GPU_MAX_HW_QUEUES, ETH_KZG_AMD_SERIAL_LANES, ETH_KZG_AMD_VERIFY_COMBINE ...)"""
import importlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.RandomState(7)
    nb = 8
    blobs = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8)
    blobs[:, :, 0] &= 0x3F
    blobs = [blobs[i].tobytes() for i in range(nb)]
    if os.environ.get("BVT_TORCH"):  # the same inside a torch process (bench.py is one): torch's streams take hardware queues too
        import torch
        torch.cuda.init()
        keep = (torch.zeros(1 << 20, device="cuda"), torch.cuda.Stream())
    ctx = kzg.DASContext(True)
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    runs = [ctx.prepare_verify_cell_kzg_proof_batch([comms[b]] * 128, list(range(128)), cells[b], proofs[b]) for b in range(nb)]
    for r in runs:
        assert r() is True
    if os.environ.get("BVT_MANY"):  # a large many-verification call first (bench.py's order): the pass slots get their big arenas
        many = [([comms[b % nb]] * 128, list(range(128)), cells[b % nb], proofs[b % nb]) for b in range(1024)]
        ver, stt = ctx.verify_cell_kzg_proof_batch_many(many)
        assert all(ver)
    for n_thr in (int(x) for x in os.environ.get("BVT_THREADS", "1,2,4,8,16,32").split(",")):
        def hammer(r):
            for _ in range(reps):
                assert r()
        ths = [threading.Thread(target=hammer, args=(runs[t % nb],)) for t in range(n_thr)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        print(f"{n_thr:3d} threads: {n_thr * reps / dt:8.0f} verifications/s  ({dt / reps * 1e3:.2f} ms per round)", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
