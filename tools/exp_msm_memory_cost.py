"""What do the MSM's table gathers cost beyond their instructions?  The same 2048-blob step with (a) random blobs -- every lane of a
wave gathers its own entry: 1.5 TB/s of random 128-B lines over the whole table -- and (b) 2048 copies of ONE blob -- the lanes of
a wave are the blobs, so a wave gathers ONE entry per addition: the same instruction stream, 1/64 of the lines.  Prints the stage
times of both (python tools/exp_msm_memory_cost.py [budget GB | max])."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ETH_KZG_AMD_TABLE_GB"] = sys.argv[1] if len(sys.argv) > 1 else "max"
import torch
kzg = importlib.import_module("rust-eth-kzg_amd")
torch.zeros(1, device="cuda")
ctx = kzg.DASContext(use_precomp=True)
n = 2048
g = torch.Generator(device="cuda").manual_seed(11)
rnd = torch.randint(0, 256, (n, 131072), dtype=torch.uint8, device="cuda", generator=g)
rnd.view(n, 4096, 32)[:, :, 0] &= 0x3F
same = rnd[:1].expand(n, 131072).contiguous()
cells = torch.empty(n * 128 * 2048, dtype=torch.uint8, device="cuda")
proofs = torch.empty(n * 128 * 48, dtype=torch.uint8, device="cuda")
print("tables: width", ctx.window_bits(), "%.1f GB" % (ctx.table_bytes() / 1e9))
for name, blobs in (("random blobs", rnd), ("2048 copies of one blob", same), ("random blobs", rnd), ("2048 copies of one blob", same)):
    run = lambda: ctx.compute_cells_and_kzg_proofs_device(n, blobs.data_ptr(), cells.data_ptr(), proofs.data_ptr(), want_status=False)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    ctx.get_stage_times()
    for _ in range(4):
        run()
    torch.cuda.synchronize()
    st = ctx.get_stage_times()
    ctx.set_profiling(False)
    print(name.ljust(26), {k: round(v[0] / 4, 2) for k, v in st.items() if v[0] > 0})
ctx.close()
