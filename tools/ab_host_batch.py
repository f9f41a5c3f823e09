"""A/B of the host-pointer batch call (eth_kzg_amd_compute_cells_and_kzg_proofs_batch, 2048 blobs) under an environment knob:
   python tools/ab_host_batch.py ETH_KZG_AMD_EARLY_MSM 0 256 512      (each value in a process of its own, alternated twice)"""
import importlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n):
    sys.path.insert(0, ROOT)
    kzg = importlib.import_module("rust-eth-kzg_amd")
    rng = np.random.RandomState(1)
    blobs = rng.randint(0, 256, size=(n, 4096, 32), dtype=np.uint8)
    blobs[:, :, 0] &= 0x3F
    blobs = np.ascontiguousarray(blobs.reshape(n, 131072))
    ctx = kzg.DASContext(True)
    while ctx.tables_ready() != 1:
        time.sleep(0.05)
    bufs = ctx.host_batch_buffers(n)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        st = ctx.compute_cells_and_kzg_proofs_batch_np(blobs, bufs)
        ts.append((time.perf_counter() - t0) * 1e3)
    assert st == [0] * n
    ts = sorted(ts[2:])
    print("%s=%s: median %.2f ms, best %.2f ms = %.0f blobs/s" % (sys.argv[2], os.environ.get(sys.argv[2]), ts[len(ts) // 2], ts[0], n / ts[len(ts) // 2] * 1e3), flush=True)
    ctx.close()


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(os.environ.get("AB_BLOBS", "2048")))
    else:
        knob, values = sys.argv[1], sys.argv[2:]
        for _ in range(2):
            for v in values:
                env = dict(os.environ)
                env[knob] = v
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", knob], env=env, check=True)
