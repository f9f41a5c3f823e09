"""Experiment: K full 2048-blob steps back to back on ONE stream vs the same K steps dealt round-robin to S streams (each call
takes its own scratch set inside the library), i.e. whole batches in flight next to each other: do the tails of one batch's
launches fill with the other's?  Usage: python tools/exp_two_batches.py [blobs] [steps]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

kzg = importlib.import_module("rust-eth-kzg_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = kzg.DASContext(True, device=0)
blobs = torch.from_numpy(bench.synth_blobs(B, 1)).to(dev)
ref_c = torch.empty(B * 128 * 2048, dtype=torch.uint8, device=dev)
ref_p = torch.empty(B * 128 * 48, dtype=torch.uint8, device=dev)
ctx.compute_cells_and_kzg_proofs_device(B, blobs.data_ptr(), ref_c.data_ptr(), ref_p.data_ptr())
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
outs = [(torch.empty_like(ref_c), torch.empty_like(ref_p)) for _ in range(3)]
for S in (1, 2, 3, 1, 2, 3):
    for rep in range(2):
        for c, p in outs:
            c.zero_(); p.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            s = k % S
            ctx.compute_cells_and_kzg_proofs_device(B, blobs.data_ptr(), outs[s][0].data_ptr(), outs[s][1].data_ptr(), want_status=False,
                                                    stream=streams[s].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    for s in range(S):
        assert torch.equal(outs[s][0], ref_c) and torch.equal(outs[s][1], ref_p)
    print(f"streams={S}: {dt / K * 1e3:.2f} ms per step -> {B * K / dt:.0f} blobs/s", flush=True)
ctx.close()
