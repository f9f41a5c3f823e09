"""Contexts constructed CONCURRENTLY on one GPU (host threads of one process): every round N threads each create a context, run one
prover call, compare the bytes and free it.  usage: python tools/probe_concurrent_ctors.py [rounds] [threads] [list]
(`list`: each thread creates a context over the device list 0,0 instead)."""
import importlib, os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ETH_KZG_AMD_TABLE_GB", "3")
import torch
torch.cuda.init()
import synth
kzg = importlib.import_module("rust-eth-kzg_amd")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_thr = int(sys.argv[2]) if len(sys.argv) > 2 else 4
use_list = len(sys.argv) > 3 and sys.argv[3] == "list"
blob = synth.seeded_blob(1)
ref_ctx = kzg.DASContext(use_precomp=True)
want = ref_ctx.compute_cells_and_kzg_proofs(blob)
ref_ctx.close()
errors = []


def work(i):
    try:
        c = kzg.DASContext(use_precomp=True, devices=[0, 0]) if use_list else kzg.DASContext(use_precomp=True, device=0)
        got = c.compute_cells_and_kzg_proofs(blob)
        if got != want:
            errors.append((i, "bytes differ"))
        c.close()
    except Exception as e:  # noqa: BLE001
        errors.append((i, repr(e)))


for r in range(rounds):
    th = [threading.Thread(target=work, args=(i,)) for i in range(n_thr)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    print("round", r, "errors", errors, flush=True)
print("probe ok" if not errors else "probe FAILED")
