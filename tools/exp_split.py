"""Experiment: one 2048-blob compute_cells_and_kzg_proofs call vs the same batch as P concurrent calls on P streams
(each takes its own scratch set inside the library): does overlapping the tails of one part with the full launches of
another pay?  Usage: python tools/exp_split.py [blobs]"""
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
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = kzg.DASContext(True, device=0)
blobs = torch.from_numpy(bench.synth_blobs(B, 1)).to(dev)
cells = torch.empty(B * 128 * 2048, dtype=torch.uint8, device=dev)
proofs = torch.empty(B * 128 * 48, dtype=torch.uint8, device=dev)
ref_c, ref_p = torch.empty_like(cells), torch.empty_like(proofs)
ctx.compute_cells_and_kzg_proofs_device(B, blobs.data_ptr(), ref_c.data_ptr(), ref_p.data_ptr())
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
for P in (1, 2, 3, 4, 2, 1):
    bounds = [B * i // P for i in range(P + 1)]
    ts = []
    for it in range(6):
        cells.zero_(); proofs.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(P):
            lo, hi = bounds[i], bounds[i + 1]
            ctx.compute_cells_and_kzg_proofs_device(hi - lo, blobs[lo:hi].data_ptr(), cells[lo * 262144:].data_ptr(), proofs[lo * 6144:].data_ptr(),
                                                    want_status=False, stream=streams[i].cuda_stream)
        torch.cuda.synchronize()
        if it >= 2:
            ts.append(time.perf_counter() - t0)
    assert torch.equal(cells, ref_c) and torch.equal(proofs, ref_p)
    ts.sort()
    print(f"P={P}: {ts[len(ts)//2]*1e3:.2f} ms  -> {B/ts[len(ts)//2]:.0f} blobs/s")
ctx.close()
