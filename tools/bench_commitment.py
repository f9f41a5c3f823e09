"""Throughput of blob_to_kzg_commitment on device-resident blobs (BASELINE config 1's operation, batched)."""
import importlib, os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
kzg = importlib.import_module("rust-eth-kzg_amd")
from oracle_lib import Oracle
ctx = kzg.DASContext(True)
for B in (1, 64, 2048):
    rng = np.random.RandomState(3)
    a = rng.randint(0, 256, size=(B, 4096, 32), dtype=np.uint8); a[:, :, 0] &= 0x3F
    d_blobs = torch.from_numpy(a.reshape(-1)).cuda()
    d_out = torch.empty(B * 48, dtype=torch.uint8, device="cuda")
    for _ in range(2): ctx.blob_to_kzg_commitment_device(B, d_blobs.data_ptr(), d_out.data_ptr(), want_status=False)
    torch.cuda.synchronize(); t = time.time(); reps = 5
    for _ in range(reps): ctx.blob_to_kzg_commitment_device(B, d_blobs.data_ptr(), d_out.data_ptr(), want_status=False)
    torch.cuda.synchronize(); dt = (time.time() - t) / reps
    if B == 1:
        assert bytes(d_out.cpu().numpy()) == Oracle(use_precomp=False, threads=1).blob_to_kzg_commitment(a[0].tobytes())
    print(json.dumps({"commitment_blobs": B, "ms": round(dt * 1e3, 3), "blobs_per_s": round(B / dt)}))
