"""Does the 2048-blob step gain from running as two (four) concurrent device-resident calls of 1024 (512) blobs on the engine's lanes
-- the tail of one call's kernels under the other's -- ?  Alternated on ONE box: python tools/probe_split_streams.py [rounds]"""
import importlib
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

kzg = importlib.import_module("rust-eth-kzg_amd")
N, STEPS = 2048, 6
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
os.environ.setdefault("ETH_KZG_AMD_TABLE_GB", "max")
torch.zeros(1, device="cuda")
ctx = kzg.DASContext(use_precomp=True)
g = torch.Generator(device="cuda").manual_seed(11)
blobs = torch.randint(0, 256, (N, 131072), dtype=torch.uint8, device="cuda", generator=g)
blobs.view(N, 4096, 32)[:, :, 0] &= 0x3F
cells = torch.empty(N * 128 * 2048, dtype=torch.uint8, device="cuda")
proofs = torch.empty(N * 128 * 48, dtype=torch.uint8, device="cuda")
ref = None


def run(parts):
    per = N // parts
    def one(k):
        ctx.compute_cells_and_kzg_proofs_device(per, blobs.data_ptr() + k * per * 131072, cells.data_ptr() + k * per * 128 * 2048,
                                                proofs.data_ptr() + k * per * 128 * 48, want_status=False)
    if parts == 1:
        one(0)
        return
    th = [threading.Thread(target=one, args=(k,)) for k in range(parts)]
    for t in th:
        t.start()
    for t in th:
        t.join()


for _ in range(3):
    run(1)
torch.cuda.synchronize()
time.sleep(8)  # (the wide tables are complete)
for r in range(rounds):
    for parts in (1, 2, 4):
        run(parts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            run(parts)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / STEPS
        digest = int(proofs.to(torch.int64).sum().item())
        ref = ref or digest
        print(f"round {r} parts={parts}: {ms:.3f} ms per 2048 blobs = {N / ms * 1e3:.0f} blobs/s  window_bits={ctx.window_bits()}  same_bytes={digest == ref}", flush=True)
ctx.close()
