"""N separate contexts on one GPU hammered by T threads with k-blob host batches (thread i uses context i mod N), results compared.
usage: python tools/probe_many_contexts.py [contexts] [threads] [blobs per batch] [seconds]"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ETH_KZG_AMD_TABLE_GB", "3")
import torch
torch.cuda.init()
import synth
kzg = importlib.import_module("rust-eth-kzg_amd")
N, T, K, S = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 4), (2, 16), (3, 16), (4, 10)))
blobs = [synth.seeded_blob(50 + i) for i in range(K)]
ctxs = [kzg.DASContext(use_precomp=True) for _ in range(N)]
want = ctxs[0].compute_cells_and_kzg_proofs_batch(blobs)
stop, errors, count = time.time() + S, [], [0] * T


def work(i):
    try:
        while time.time() < stop and not errors:
            if ctxs[i % N].compute_cells_and_kzg_proofs_batch(blobs) != want:
                errors.append((i, "bytes differ"))
            count[i] += 1
    except BaseException as e:  # noqa: BLE001
        errors.append((i, repr(e)))


th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
for t in th:
    t.start()
for t in th:
    t.join()
print(f"{N} contexts, {T} threads, batches of {K}: {sum(count)} calls, errors {errors[:3]}")
print("probe ok" if not errors else "probe FAILED")
for c in ctxs:
    c.close()
