"""Stage times of compute_cells_and_kzg_proofs on device-resident blobs at several batch sizes (the small and middle batches of
BASELINE configs 4 and 5): python tools/stage_times_by_batch.py 32 64 128 256 512"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ETH_KZG_AMD_TABLE_GB", "max")
import torch
kzg = importlib.import_module("rust-eth-kzg_amd")
sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256, 512]
torch.zeros(1, device="cuda")
ctx = kzg.DASContext(use_precomp=True)
g = torch.Generator(device="cuda").manual_seed(7)
for n in sizes:
    blobs = torch.randint(0, 256, (n, 131072), dtype=torch.uint8, device="cuda", generator=g)
    blobs.view(n, 4096, 32)[:, :, 0] &= 0x3F
    cells = torch.empty(n * 128 * 2048, dtype=torch.uint8, device="cuda")
    proofs = torch.empty(n * 128 * 48, dtype=torch.uint8, device="cuda")
    run = lambda: ctx.compute_cells_and_kzg_proofs_device(n, blobs.data_ptr(), cells.data_ptr(), proofs.data_ptr(), want_status=False)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / 10 * 1e3
    ctx.set_profiling(True)
    ctx.get_stage_times()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    st = ctx.get_stage_times()
    ctx.set_profiling(False)
    print(n, "blobs: %.3f ms per call;" % plain, {k: round(v[0] / 5, 3) for k, v in st.items() if v[0] > 0})
ctx.close()
