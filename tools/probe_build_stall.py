"""A caller on the start tables while the helper thread allocates and builds the wide ones: one single-blob call every 5 ms,
calls slower than 50 ms reported with the builder's trace lines on the same clock (ETH_KZG_AMD_TRACE=1 ETH_KZG_AMD_TRACE=50)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")
t0 = time.perf_counter()
ctx = kzg.DASContext(True, wait_tables=False)
blob = bytes(131072)
first = ctx.compute_cells_and_kzg_proofs(blob)
print(f"first result at {time.perf_counter() - t0:.2f} s", flush=True)
n, worst = 0, 0.0
while ctx.tables_ready(0) == 0:
    t1 = time.perf_counter()
    assert ctx.compute_cells_and_kzg_proofs(blob) == first
    dt = time.perf_counter() - t1
    n += 1
    if dt > 0.05:
        print(f"call {n} at {t1 - t0:.2f} s took {dt * 1e3:.0f} ms (groups ready {ctx.table_groups_ready()})", flush=True)
    worst = max(worst, dt)
    time.sleep(0.005)
print(f"{n} calls during the build, longest {worst * 1e3:.0f} ms; wide tables after {time.perf_counter() - t0:.2f} s", flush=True)
ctx.close()
