"""Debug driver: N threads calling eth_kzg_verify_cell_kzg_proof_batch on one context (the combiner path)."""
import faulthandler, importlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
faulthandler.dump_traceback_later(40, exit=True)
import synth
kzg = importlib.import_module("rust-eth-kzg_amd")
ctx = kzg.DASContext(True)
blobs = [synth.seeded_blob(i) for i in range(2)]
st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
runs = [ctx.prepare_verify_cell_kzg_proof_batch([comms[b]] * 128, list(range(128)), cells[b], proofs[b]) for b in range(2)]
print("single:", runs[0](), flush=True)
N, R = int(sys.argv[1]), int(sys.argv[2])
def w(t):
    for r in range(R):
        assert runs[(t + r) % 2]()
t0 = time.time()
ths = [threading.Thread(target=w, args=(t,)) for t in range(N)]
[t.start() for t in ths]; [t.join() for t in ths]
print("ok", N * R / (time.time() - t0), "per s", flush=True)
os._exit(0)
