import importlib, os, sys, threading, time
import numpy as np
sys.path.insert(0, "/root/repo")
kzg = importlib.import_module("rust-eth-kzg_amd")
rng = np.random.RandomState(7)
nb = 64
blobs = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8)
blobs[:, :, 0] &= 0x3F
blobs = [blobs[i].tobytes() for i in range(nb)]
ctx = kzg.DASContext(True)
st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
probs = [([comms[b]] * 128, list(range(128)), cells[b], proofs[b]) for b in range(nb)]
def threads_test(tag):
    for n_thr in (4, 32):
        runs_t = [ctx.prepare_verify_cell_kzg_proof_batch(*probs[b % nb]) for b in range(n_thr)]
        reps = 25
        def hammer(r):
            for _ in range(reps):
                assert r()
        for r in runs_t[:2]:
            r()
        ths = [threading.Thread(target=hammer, args=(r,)) for r in runs_t]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        print(tag, n_thr, "threads:", round(n_thr * reps / (time.perf_counter() - t0)), flush=True)
threads_test("fresh")
threads_test("again")
many = [probs[j % nb] for j in range(1024)]
run_many = ctx.prepare_verify_cell_kzg_proof_batch_many(many)
run_many(); run_many()
threads_test("after verify_many(1024)")
threads_test("again")
ctx.close()
