"""Soak: N host threads on ONE context, each picking operations at random for a fixed time -- single verifications (valid and
tampered), small many-verification calls, single-blob and 64-blob prover calls, single-blob recovery -- every result compared with
the bytes / verdicts computed up front on one thread.  usage: python tools/soak_mixed_threads.py [seconds] [threads]"""
import importlib, os, random, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    n_thr = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rng = np.random.RandomState(11)
    nb = 64
    blobs = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8)
    blobs[:, :, 0] &= 0x3F
    blobs = [blobs[i].tobytes() for i in range(nb)]
    ctx = kzg.DASContext(True, wait_tables=False)  # the soak starts while the wide tables are still being built
    # SOAK_CONTEXTS=N: N separate contexts on the GPU instead of one (thread i works on context i mod N)
    extra = [kzg.DASContext(True, wait_tables=False) for _ in range(int(os.environ.get("SOAK_CONTEXTS", "1")) - 1)]
    all_ctx = [ctx] + extra
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    assert st == [0] * nb
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    probs = [([comms[b]] * 128, list(range(128)), cells[b], proofs[b]) for b in range(8)]
    bad = []
    for b in range(8):
        p = list(proofs[b]); p[5] = proofs[(b + 1) % nb][5]
        bad.append(([comms[b]] * 128, list(range(128)), cells[b], p))
    ops = os.environ.get("SOAK_OPS", "verify,verify,verify_bad,many,compute1,compute64,recover,commit,invalid").split(",")  # SOAK_OPS: a sub-list (bisecting)
    stop = time.time() + seconds
    counts, errors, lock = {}, [], threading.Lock()

    def worker(seed):
        r = random.Random(seed)
        mine = {}
        ctx = all_ctx[seed % len(all_ctx)]
        try:
            while time.time() < stop and not errors:
                op = r.choice(ops)
                b = r.randrange(8)
                if op == "verify":
                    assert ctx.verify_cell_kzg_proof_batch(*probs[b]) is True
                elif op == "verify_bad":
                    assert ctx.verify_cell_kzg_proof_batch(*bad[b]) is False
                elif op == "many":
                    k = r.randrange(1, 6)
                    sel = [(r.randrange(8), r.random() < 0.3) for _ in range(k)]
                    ver, stt = ctx.verify_cell_kzg_proof_batch_many([bad[i] if w else probs[i] for i, w in sel])
                    assert stt == [0] * k and ver == [not w for _, w in sel]
                elif op == "compute1":
                    c, p = ctx.compute_cells_and_kzg_proofs(blobs[b])
                    assert c == cells[b] and p == proofs[b]
                elif op == "compute64":
                    s2, c2, p2 = ctx.compute_cells_and_kzg_proofs_batch(blobs)
                    assert s2 == [0] * nb and c2[b] == cells[b] and p2[nb - 1] == proofs[nb - 1]
                elif op.startswith("computeN"):  # computeN<k>: a host-pointer batch of k blobs (bisecting a batch-size regime)
                    k = int(op[8:])
                    s2, c2, p2 = ctx.compute_cells_and_kzg_proofs_batch(blobs[:k])
                    assert s2 == [0] * k and c2[b % k] == cells[b % k] and p2[k - 1] == proofs[k - 1]
                elif op == "invalid":  # error paths under concurrency: every one of these must raise, and nothing around them may notice
                    which = r.randrange(4)
                    try:
                        if which == 0:
                            ctx.compute_cells_and_kzg_proofs(b"\xff" * 32 + blobs[b][32:])
                        elif which == 1:
                            ctx.recover_cells_and_kzg_proofs(list(range(63)), [cells[b][i] for i in range(63)])
                        elif which == 2:
                            ctx.verify_cell_kzg_proof_batch([comms[b]], [128], [cells[b][0]], [proofs[b][0]])
                        else:
                            ctx.blob_to_kzg_commitment(blobs[b][:-32] + b"\xff" * 32)
                        raise AssertionError("an invalid input was accepted")
                    except kzg.KzgError:
                        pass
                elif op == "recover":
                    idx = sorted(r.sample(range(128), 64))
                    c, p = ctx.recover_cells_and_kzg_proofs(idx, [cells[b][i] for i in idx])
                    assert c == cells[b] and p == proofs[b]
                else:
                    assert ctx.blob_to_kzg_commitment(blobs[b]) == comms[b]
                mine[op] = mine.get(op, 0) + 1
        except BaseException as e:  # noqa: BLE001 -- any failure ends the soak and is reported
            errors.append(repr(e))
        with lock:
            for k, v in mine.items():
                counts[k] = counts.get(k, 0) + v

    churned = [0]

    def churn():  # SOAK_CHURN=1: contexts created and freed beside the workers (some freed while their wide tables are being built)
        r = random.Random(7)
        try:
            while time.time() < stop and not errors:
                c = kzg.DASContext(True, wait_tables=r.random() < 0.3)
                if r.random() < 0.5:
                    assert c.blob_to_kzg_commitment(blobs[0]) == comms[0]
                time.sleep(r.random() * 0.5)
                c.close()
                churned[0] += 1
        except BaseException as e:  # noqa: BLE001
            errors.append("churn: " + repr(e))

    ths = [threading.Thread(target=worker, args=(1000 + i,)) for i in range(n_thr)]
    if os.environ.get("SOAK_CHURN") == "1":
        ths.append(threading.Thread(target=churn))
    t0 = time.time()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    print(f"{n_thr} threads, {time.time() - t0:.1f} s, tables_ready={ctx.tables_ready()}: " + ", ".join(f"{k} {v}" for k, v in sorted(counts.items())) + (f", contexts churned {churned[0]}" if churned[0] else ""))
    if errors:
        print("FAILED:", errors[:3])
        sys.exit(1)
    print("soak ok")
    for c in all_ctx:
        c.close()


if __name__ == "__main__":
    main()
