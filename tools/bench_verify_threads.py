"""Single eth_kzg_verify_cell_kzg_proof_batch calls (the reference bench's shape: 128 cells, one commitment) from N host threads
on ONE context -- the reference's usage model (bindings/node/src/lib.rs:92-299): verifications per second at N = 1, 2, 4, 8, 16, 32.
usage: python tools/bench_verify_threads.py [reps]   (env: GPU_MAX_HW_QUEUES, ETH_KZG_AMD_SERIAL_LANES, ETH_KZG_AMD_VERIFY_COMBINE ...)"""
import importlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.RandomState(7)
    nb = 8
    blobs = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8)
    blobs[:, :, 0] &= 0x3F
    blobs = [blobs[i].tobytes() for i in range(nb)]
    if os.environ.get("BVT_TORCH"):  # the same inside a torch process (bench.py is one): torch's streams take hardware queues too
        import torch
        torch.cuda.init()
        keep = (torch.zeros(1 << 20, device="cuda"), torch.cuda.Stream())
    if os.environ.get("BVT_TORCH_LATE"):  # bench.py's order: the context first (torch imported, device set, nothing launched), torch's streams after it
        import torch
        torch.cuda.set_device(0)
    ctx = kzg.DASContext(True)
    if os.environ.get("BVT_TORCH_LATE"):
        keep = (torch.zeros(1 << 20, device="cuda"), torch.cuda.Stream())
        with torch.cuda.stream(keep[1]):
            keep[0].add_(1)
        torch.cuda.synchronize()
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    runs = [ctx.prepare_verify_cell_kzg_proof_batch([comms[b]] * 128, list(range(128)), cells[b], proofs[b]) for b in range(nb)]
    for r in runs:
        assert r() is True
    if os.environ.get("BVT_MANY"):  # a large many-verification call first (bench.py's order): the pass slots get their big arenas
        many = [([comms[b % nb]] * 128, list(range(128)), cells[b % nb], proofs[b % nb]) for b in range(1024)]
        ver, stt = ctx.verify_cell_kzg_proof_batch_many(many)
        assert all(ver)
    steps = os.environ.get("BVT_STEPS", "").split(",")  # what bench.py does to the context before its thread figures, piece by piece
    if "dev2048" in steps or "abi2" in steps:
        big = rng.randint(0, 256, size=(2048, 4096, 32), dtype=np.uint8)
        big[:, :, 0] &= 0x3F
        big = np.ascontiguousarray(big.reshape(2048, 131072))
    if "dev2048" in steps:
        import torch
        stream = torch.cuda.Stream()
        d_b = torch.from_numpy(big).cuda()
        d_c = torch.empty(2048 * 128 * 2048, dtype=torch.uint8, device="cuda")
        d_p = torch.empty(2048 * 128 * 48, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            with torch.cuda.stream(stream):
                ctx.compute_cells_and_kzg_proofs_device(2048, d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr(), want_status=False, stream=stream.cuda_stream)
            torch.cuda.synchronize()
    if "abi2" in steps:
        bufs, bufs2 = ctx.host_batch_buffers(2048), ctx.host_batch_buffers(2048)
        tt = [threading.Thread(target=lambda bf=bf: [ctx.compute_cells_and_kzg_proofs_batch_np(big, bf) for _ in range(2)]) for bf in (bufs, bufs2)]
        for t in tt:
            t.start()
        for t in tt:
            t.join()
    if "recover" in steps:
        for _ in range(3):
            ctx.recover_cells_and_kzg_proofs(list(range(64)), cells[0][:64])
    if "config3" in steps:
        C, I, L, P = [], [], [], []
        for b in range(nb):
            for k in range(128):
                C.append(comms[b]); I.append(k); L.append(cells[b][k]); P.append(proofs[b][k])
        for _ in range(3):
            assert ctx.verify_cell_kzg_proof_batch(C, I, L, P)
    if "manybad" in steps:
        many = [([comms[b % nb]] * 128, list(range(128)), cells[b % nb], proofs[b % nb]) for b in range(1024)]
        for bad_at in ([77], [3, 77, 200, 201, 512, 700, 901, 1023]):
            mb = list(many)
            for j in bad_at:
                pj = list(many[j][3]); pj[9] = many[(j + 1) % nb][3][9]
                mb[j] = (many[j][0], many[j][1], many[j][2], pj)
            for _ in range(3):
                ver, stt = ctx.verify_cell_kzg_proof_batch_many(mb)
            assert ver == [j not in bad_at for j in range(1024)]
    if "vdev" in steps or "recoverdev" in steps:
        import torch
        flat = np.frombuffer(b"".join(blobs), dtype=np.uint8)
        d_b = torch.from_numpy(flat.copy()).cuda()
        d_c = torch.empty(nb * 128 * 2048, dtype=torch.uint8, device="cuda")
        d_p = torch.empty(nb * 128 * 48, dtype=torch.uint8, device="cuda")
        ctx.compute_cells_and_kzg_proofs_device(nb, d_b.data_ptr(), d_c.data_ptr(), d_p.data_ptr())
        torch.cuda.synchronize()
    if "vdev" in steps:
        d_comm = torch.frombuffer(bytearray(b"".join(comms)), dtype=torch.uint8).cuda().view(nb, 1, 48).expand(nb, 128, 48).contiguous().view(-1)
        d_idx = torch.arange(128, dtype=torch.int64, device="cuda").repeat(nb)
        torch.cuda.synchronize()
        for _ in range(4):
            assert ctx.verify_cell_kzg_proof_batch_device(nb * 128, d_comm.data_ptr(), d_idx.data_ptr(), d_c.data_ptr(), d_p.data_ptr()) is True
    if "recoverdev" in steps:
        erased = d_c.view(nb, 128, 2048).clone()
        erased[:, 1::2, :] = 0xFF
        d_oc, d_op = torch.empty_like(d_c), torch.empty_like(d_p)
        for _ in range(3):
            stt = ctx.recover_cells_and_kzg_proofs_device(nb, erased.data_ptr(), [list(range(0, 128, 2))] * nb, d_oc.data_ptr(), d_op.data_ptr())
            torch.cuda.synchronize()
        assert stt == [0] * nb
    for n_thr in (int(x) for x in os.environ.get("BVT_THREADS", "1,2,4,8,16,32").split(",")):
        def hammer(r):
            for _ in range(reps):
                assert r()
        ths = [threading.Thread(target=hammer, args=(runs[t % nb],)) for t in range(n_thr)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        print(f"{n_thr:3d} threads: {n_thr * reps / dt:8.0f} verifications/s  ({dt / reps * 1e3:.2f} ms per round)", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
