"""Host-pointer ABI throughput: eth_kzg_amd_compute_cells_and_kzg_proofs_batch on host buffers (what a C / Go / Java
caller hands over), next to the device-resident entry point on the same blobs; also the single-call latency of
eth_kzg_compute_cells_and_kzg_proofs and N threads calling it concurrently on one context."""
import concurrent.futures as cf
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    rng = np.random.RandomState(1)
    blobs = rng.randint(0, 256, size=(n, 4096, 32), dtype=np.uint8)
    blobs[:, :, 0] &= 0x3F
    blobs = np.ascontiguousarray(blobs.reshape(n, 131072))
    ctx = kzg.DASContext(True)
    bufs = ctx.host_batch_buffers(n)
    for _ in range(2):
        st = ctx.compute_cells_and_kzg_proofs_batch_np(blobs, bufs)
    assert st == [0] * n
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.compute_cells_and_kzg_proofs_batch_np(blobs, bufs)
        t.append(time.perf_counter() - t0)
    host = n / min(t)
    d_blobs = torch.from_numpy(blobs.reshape(-1)).cuda()
    d_cells = torch.empty(n * 128 * 2048, dtype=torch.uint8, device="cuda")
    d_proofs = torch.empty(n * 128 * 48, dtype=torch.uint8, device="cuda")
    td = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.compute_cells_and_kzg_proofs_device(n, d_blobs.data_ptr(), d_cells.data_ptr(), d_proofs.data_ptr(), want_status=False)
        torch.cuda.synchronize()
        td.append(time.perf_counter() - t0)
    dev = n / min(td)
    same = np.array_equal(d_cells.cpu().numpy().reshape(n, 128, 2048), bufs["cells"]) and \
        np.array_equal(d_proofs.cpu().numpy().reshape(n, 128, 48), bufs["proofs"])
    print(f"n={n}: host-pointer batch {host:.0f} blobs/s ({min(t)*1e3:.1f} ms; all runs {[round(x*1e3,1) for x in t]}), "
          f"device-resident {dev:.0f} blobs/s ({min(td)*1e3:.1f} ms), ratio {host/dev:.3f}, outputs identical: {same}")
    # single-call latency through the reference's own entry point
    one = blobs[0].tobytes()
    ctx.compute_cells_and_kzg_proofs(one)
    lat = []
    for _ in range(10):
        t0 = time.perf_counter()
        ctx.compute_cells_and_kzg_proofs(one)
        lat.append(time.perf_counter() - t0)
    print(f"eth_kzg_compute_cells_and_kzg_proofs single call: {min(lat)*1e3:.2f} ms (python wrapper included)")
    # concurrency: T threads x K single-blob calls on ONE context
    for T in (1, 2, 4, 8):
        K = 16
        def work(i):
            for _ in range(K):
                ctx.compute_cells_and_kzg_proofs(one)
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(max_workers=T) as ex:
            list(ex.map(work, range(T)))
        dt = time.perf_counter() - t0
        print(f"  {T} threads x {K} single-blob calls: {T*K/dt:.0f} blobs/s")
    ctx.close()


if __name__ == "__main__":
    main()
