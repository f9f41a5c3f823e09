import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
kzg = importlib.import_module("rust-eth-kzg_amd")
rng = np.random.RandomState(3)
nb = 8
blobs = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8)
blobs[:, :, 0] &= 0x3F
blobs = [blobs[i].tobytes() for i in range(nb)]
ctx = kzg.DASContext(True)
st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
_, comms = ctx.blob_to_kzg_commitment_batch(blobs)
probs = [([comms[b]] * 128, list(range(128)), cells[b], proofs[b]) for b in range(nb)]
for B in (2, 4, 8, 16, 32):
    many = [probs[j % nb] for j in range(B)]
    bad = list(many[1][3]); bad[3] = many[0][3][3] if many[0][3][3] != many[1][3][3] else many[0][3][4]
    manyb = list(many); manyb[1] = (many[1][0], many[1][1], many[1][2], bad)
    run = ctx.prepare_verify_cell_kzg_proof_batch_many(many)
    runb = ctx.prepare_verify_cell_kzg_proof_batch_many(manyb)
    ver, stt = runb()
    assert stt == [0] * B and ver == [j != 1 for j in range(B)], (ver, stt)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter()
        ver, stt = run()
        ts.append((time.perf_counter() - t0) * 1e3)
        assert all(ver)
    ts.sort()
    print("COOP=%s pass of %2d problems: median %.2f ms" % (os.environ.get("ETH_KZG_AMD_COOP_POINTS"), B, ts[20]))
ctx.close()
