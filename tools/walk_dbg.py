"""Debug driver of the linear map's ticket walker: device-resident path, small batch."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
torch.cuda.init()
kzg = importlib.import_module("rust-eth-kzg_amd")
import bench
n = int(sys.argv[1])
ctx = kzg.DASContext(True)
blobs = torch.from_numpy(bench.synth_blobs(n, 3)).cuda()
cells = torch.empty(n * 262144, dtype=torch.uint8, device="cuda"); proofs = torch.empty(n * 6144, dtype=torch.uint8, device="cuda")
t0 = time.time()
st = ctx.compute_cells_and_kzg_proofs_device(n, blobs.data_ptr(), cells.data_ptr(), proofs.data_ptr())
print("done", n, round(time.time() - t0, 3), st[:3], flush=True)
os.environ["ETH_KZG_AMD_SLP_WALK"] = "0"
c2 = kzg.DASContext(True)
cells2, proofs2 = torch.empty_like(cells), torch.empty_like(proofs)
c2.compute_cells_and_kzg_proofs_device(n, blobs.data_ptr(), cells2.data_ptr(), proofs2.data_ptr())
print("equal:", torch.equal(proofs, proofs2), torch.equal(cells, cells2), flush=True)
os._exit(0)
