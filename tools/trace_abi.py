"""Three calls of eth_kzg_amd_compute_cells_and_kzg_proofs_batch on 2048 host blobs, for rocprofv3 --kernel-trace --memory-copy-trace."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.RandomState(1)
blobs = rng.randint(0, 256, size=(n, 4096, 32), dtype=np.uint8)
blobs[:, :, 0] &= 0x3F
blobs = np.ascontiguousarray(blobs.reshape(n, 131072))
ctx = kzg.DASContext(True)
bufs = ctx.host_batch_buffers(n)
for _ in range(3):
    st = ctx.compute_cells_and_kzg_proofs_batch_np(blobs, bufs)
assert st == [0] * n
ctx.close()
