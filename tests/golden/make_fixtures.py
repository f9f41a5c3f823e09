#!/usr/bin/env python3
"""Convert the reference's consensus-spec KZG test vectors into compact fixtures.

Source (read only in the build container, never at test time):
    /root/reference/test_vectors/<family>/kzg-mainnet/<case>/data.y{a,}ml
families: blob_to_kzg_commitment, compute_cells_and_kzg_proofs,
          verify_cell_kzg_proof_batch, recover_cells_and_kzg_proofs,
          compute_kzg_proof, compute_blob_kzg_proof, verify_kzg_proof, verify_blob_kzg_proof,
          verify_blob_kzg_proof_batch (crates/eip4844/tests/*.rs)
(the vectors the reference's own integration tests replay:
 crates/eip7594/tests/{blob_to_kzg_commitment,compute_cells_and_kzg_proofs,
 verify_cell_kzg_proof_batch,recover_cells_and_kzg_proofs}.rs)

Output (committed, pure data = inputs + expected outputs):
    tests/golden/vectors.json     case structure; every byte string replaced by {"$b": <pool index>}
    tests/golden/pool.bin.xz      LZMA of the concatenated unique byte strings
    tests/golden/pool_index.json  [offset, length] per pool entry

Also converts the trusted setup (public ceremony data the reference embeds,
crates/trusted_setup/data/trusted_setup_4096.json) to a flat binary the product loads:
    rust-eth-kzg_amd/data/trusted_setup_4096.bin
      = "KZGSRS01" | u32le n_g1 | u32le n_g2 | n_g1 x 48 B compressed G1 (monomial) | n_g2 x 96 B compressed G2 (monomial)
"""
import glob, json, lzma, os, struct, sys
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FAMILIES = ["blob_to_kzg_commitment", "compute_cells_and_kzg_proofs",
            "verify_cell_kzg_proof_batch", "recover_cells_and_kzg_proofs",
            # EIP-4844 single-point families
            "compute_kzg_proof", "compute_blob_kzg_proof", "verify_kzg_proof", "verify_blob_kzg_proof",
            "verify_blob_kzg_proof_batch"]

pool, pool_ix = [], {}

def intern(b: bytes):
    if b not in pool_ix:
        pool_ix[b] = len(pool)
        pool.append(b)
    return {"$b": pool_ix[b]}

def conv(v):
    if isinstance(v, str):
        assert v.startswith("0x"), v[:20]
        return intern(bytes.fromhex(v[2:]))
    if isinstance(v, (list, tuple)):
        return [conv(x) for x in v]
    if isinstance(v, dict):
        return {k: conv(x) for k, x in v.items()}
    return v  # int / bool / None

def main():
    out = {}
    for fam in FAMILIES:
        cases = {}
        for d in sorted(glob.glob(f"{REF}/test_vectors/{fam}/kzg-mainnet/*")):
            files = glob.glob(d + "/data.y*ml")
            assert len(files) == 1, d
            y = yaml.safe_load(open(files[0]))
            name = os.path.basename(d)
            prefix = fam + "_case_"
            assert name.startswith(prefix)
            cases[name[len(prefix):]] = {"input": conv(y["input"]), "output": conv(y["output"])}
        out[fam] = cases
        print(fam, len(cases), "cases")
    blob = b"".join(pool)
    index, off = [], 0
    for b in pool:
        index.append([off, len(b)]); off += len(b)
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    with open(os.path.join(HERE, "pool_index.json"), "w") as f:
        json.dump(index, f, separators=(",", ":"))
    with open(os.path.join(HERE, "pool.bin.xz"), "wb") as f:
        f.write(lzma.compress(blob, preset=9 | lzma.PRESET_EXTREME))
    print("pool entries", len(pool), "raw bytes", len(blob))

    ts = json.load(open(f"{REF}/crates/trusted_setup/data/trusted_setup_4096.json"))
    g1 = [bytes.fromhex(h[2:]) for h in ts["g1_monomial"]]
    g2 = [bytes.fromhex(h[2:]) for h in ts["g2_monomial"]]
    assert len(g1) == 4096 and len(g2) == 65
    assert all(len(x) == 48 for x in g1) and all(len(x) == 96 for x in g2)
    dst = os.path.join(ROOT, "rust-eth-kzg_amd", "data", "trusted_setup_4096.bin")
    with open(dst, "wb") as f:
        f.write(b"KZGSRS01" + struct.pack("<II", len(g1), len(g2)) + b"".join(g1) + b"".join(g2))
    print("trusted setup ->", dst, os.path.getsize(dst), "bytes")

if __name__ == "__main__":
    main()
