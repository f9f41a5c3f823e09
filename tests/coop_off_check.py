"""Run by tests/test_gpu_parity.py in a process of its own with ETH_KZG_AMD_COOP_POINTS=0 (the limit is read once per process):
the one-lane-per-point forms of the verification kernels -- which small inputs no longer reach by default -- on the
single path, a small many-verification pass and the EIP-4844 verifier, valid and tampered."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kzg = importlib.import_module("rust-eth-kzg_amd")


def main():
    rng = np.random.RandomState(5)
    nb = 3
    blobs = rng.randint(0, 256, size=(nb, 4096, 32), dtype=np.uint8)
    blobs[:, :, 0] &= 0x3F
    blobs = [blobs[i].tobytes() for i in range(nb)]
    ctx = kzg.DASContext(True)
    st, cells, proofs = ctx.compute_cells_and_kzg_proofs_batch(blobs)
    assert st == [0] * nb
    _, comms = ctx.blob_to_kzg_commitment_batch(blobs)
    probs = [([comms[b]] * 128, list(range(128)), cells[b], proofs[b]) for b in range(nb)]
    bad = list(proofs[1]); bad[7] = proofs[0][7]
    tampered = ([comms[1]] * 128, list(range(128)), cells[1], bad)
    assert ctx.verify_cell_kzg_proof_batch(*probs[0]) is True
    assert ctx.verify_cell_kzg_proof_batch(*tampered) is False
    ver, stt = ctx.verify_cell_kzg_proof_batch_many([probs[0], tampered, probs[2]])
    assert stt == [0, 0, 0] and ver == [True, False, True], (ver, stt)
    # a proof that is a curve point OUTSIDE the subgroup (the oracle confirms: on the curve, rejected with the subgroup test): an error
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib
    x = 5
    while True:
        cand = bytearray(x.to_bytes(48, "big")); cand[0] |= 0x80
        if oracle_lib.g1_validate(bytes(cand), False) == 0 and oracle_lib.g1_validate(bytes(cand), True) != 0:
            break
        x += 1
    p2 = list(proofs[0]); p2[3] = bytes(cand)
    try:
        ctx.verify_cell_kzg_proof_batch([comms[0]] * 128, list(range(128)), cells[0], p2)
        raise AssertionError("a proof outside the subgroup was accepted as input")
    except kzg.KzgError:
        pass
    ver, stt = ctx.verify_cell_kzg_proof_batch_many([probs[0], ([comms[0]] * 128, list(range(128)), cells[0], p2)])
    assert stt[0] == 0 and ver[0] is True and stt[1] != 0, (ver, stt)
    z = (12345).to_bytes(32, "big")
    proof, y = ctx.compute_kzg_proof(blobs[0], z)
    assert ctx.verify_kzg_proof(comms[0], z, y, proof) is True
    assert ctx.verify_kzg_proof(comms[1], z, y, proof) is False
    ctx.close()
    print("coop-off ok")


if __name__ == "__main__":
    main()
