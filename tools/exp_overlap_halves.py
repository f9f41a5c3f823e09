"""Round 4 experiment: the lane groups of a batch in two halves, half A's linear map on the work set's second stream
next to half B's MSM (ETH_KZG_AMD_OVERLAP_HALVES=1) against the plain order: ms per call at 128 ... 1024 blobs and a hash of the proofs."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
kzg = importlib.import_module("rust-eth-kzg_amd")
rng = np.random.RandomState(3)
ctx = kzg.DASContext(True)
stream = torch.cuda.Stream()
for n in (128, 256, 384, 512, 1024):
    blobs = rng.randint(0, 256, size=(n, 4096, 32), dtype=np.uint8); blobs[:, :, 0] &= 0x3F
    d_b = torch.from_numpy(blobs.reshape(-1)).cuda()
    d_c = torch.empty(n * 128 * 2048, dtype=torch.uint8, device="cuda"); d_p = torch.empty(n * 128 * 48, dtype=torch.uint8, device="cuda")
    ts = []
    for it in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            ctx.compute_cells_and_kzg_proofs_device(n, d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), want_status=False, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        if it >= 2: ts.append(time.perf_counter() - t0)
    ts.sort()
    import hashlib
    print(n, round(ts[len(ts)//2]*1e3, 2), "ms", hashlib.sha256(d_p.cpu().numpy().tobytes()).hexdigest()[:12], flush=True)
ctx.close()
