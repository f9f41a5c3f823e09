"""First GPU bring-up script: context timing + golden vectors, verbose."""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import vectors
kzg = importlib.import_module("rust-eth-kzg_amd")
t = time.time(); ctx = kzg.DASContext(True); print("context: %.2fs table %.2f GB c=%d" % (time.time() - t, ctx.table_bytes() / 1e9, ctx.window_bits()), flush=True)
for fam, fn in (("blob_to_kzg_commitment", ctx.blob_to_kzg_commitment),):
    for name, case in sorted(vectors.load(fam).items()):
        t = time.time()
        try: out = fn(case["input"]["blob"])
        except kzg.KzgError as e: out = None
        print(fam, name, out == case["output"], "%.3fs" % (time.time() - t), flush=True)
for name, case in sorted(vectors.load("compute_cells_and_kzg_proofs").items()):
    t = time.time()
    try: out = ctx.compute_cells_and_kzg_proofs(case["input"]["blob"])
    except kzg.KzgError as e: out = None
    exp = case["output"]
    if exp is None: print(name, out is None)
    else:
        print(name, "cells", out[0] == exp[0], "proofs", out[1] == exp[1], sum(a == b for a, b in zip(out[1], exp[1])), "%.3fs" % (time.time() - t), flush=True)
