"""The drop-in boundary from Node: bindings/node is an N-API addon over include/c_eth_kzg.h with the surface of the reference's
Node binding (bindings/node/index.d.ts: DasContextJs, sync and async methods).  CPU: it builds and loads; GPU: it reproduces
a golden vector through every method, with eight async calls in flight on one context."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None or not os.path.exists("/usr/include/node/node_api.h"), reason="no node / N-API headers")


def _build():
    assert os.path.exists(os.path.join(ROOT, "rust-eth-kzg_amd", "libc_eth_kzg.so")), "build the HIP extension first"
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "bindings", "node")])


def test_addon_builds_and_exports_the_reference_surface():
    _build()
    code = ("const k = require(%r); const m = Object.getOwnPropertyNames(k.DasContextJs.prototype).sort();"
            "console.log(JSON.stringify({m, c: [k.BYTES_PER_BLOB, k.BYTES_PER_CELL, k.BYTES_PER_PROOF, k.MAX_NUM_COLUMNS], s: typeof k.DasContextJs.create}))"
            % os.path.join(ROOT, "bindings", "node"))
    out = json.loads(subprocess.check_output([NODE, "-e", code], text=True))
    want = ["blobToKzgCommitment", "computeCells", "computeCellsAndKzgProofs", "recoverCellsAndKzgProofs", "verifyCellKzgProofBatch",
            "computeKzgProof", "computeBlobKzgProof", "verifyKzgProof", "verifyBlobKzgProof", "verifyBlobKzgProofBatch"]
    for name in want:
        assert name in out["m"] and "async" + name[0].upper() + name[1:] in out["m"], name
    assert out["c"] == [131072, 2048, 48, 128] and out["s"] == "function"


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0,0"], ids=["one-device", "device-list-0,0"])
def test_node_reproduces_the_golden_vector(tmp_path, devices):
    """devices = "0,0": the same UNCHANGED Node host with ETH_KZG_AMD_DEVICES set -- its one DasContextJs then spans a device list (two
    engines on the test box's GPU; bindings/node/src/lib.rs:35,75 shares one context between all its async calls), and every method
    still reproduces the vector: how a reference host uses all GPUs of a node without a line of change."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import vectors
    case = vectors.load("compute_cells_and_kzg_proofs")["valid_4aedd1a2a3933c3e"]
    comm = [c["output"] for c in vectors.load("blob_to_kzg_commitment").values() if c["input"]["blob"] == case["input"]["blob"]][0]
    _build()
    blob_path, want_path = tmp_path / "blob.bin", tmp_path / "want.bin"
    blob_path.write_bytes(case["input"]["blob"])
    want_path.write_bytes(b"".join(case["output"][0]) + b"".join(case["output"][1]) + comm)
    env = dict(os.environ)
    env.pop("ETH_KZG_AMD_DEVICES", None)
    if devices:
        env["ETH_KZG_AMD_DEVICES"] = devices
        env["ETH_KZG_AMD_TABLE_GB"] = "22"  # (two engines on one GPU share these tables; small so that the module stays quick)
    p = subprocess.run([NODE, os.path.join(ROOT, "tests", "node", "spec.js"), str(blob_path), str(want_path)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out and all(out.values()), out
