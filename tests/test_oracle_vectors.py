"""Pin the CPU oracle (oracle/liboracle.so) against the reference's own golden vectors.

Mirrors the reference's integration tests
(crates/eip7594/tests/{blob_to_kzg_commitment,compute_cells_and_kzg_proofs,
verify_cell_kzg_proof_batch,recover_cells_and_kzg_proofs}.rs): every case must reproduce
the expected bytes exactly; error cases map to `output: null`; an invalid proof maps to False.
"""
import pytest

import vectors
from oracle_lib import OracleError


def _call(fn, *a):
    try:
        return fn(*a)
    except OracleError:
        return None


@pytest.mark.parametrize("name,case", sorted(vectors.load("blob_to_kzg_commitment").items()))
def test_blob_to_kzg_commitment(oracle, name, case):
    assert _call(oracle.blob_to_kzg_commitment, case["input"]["blob"]) == case["output"]


@pytest.mark.parametrize("precomp", [True, False])
@pytest.mark.parametrize("name,case", sorted(vectors.load("compute_cells_and_kzg_proofs").items()))
def test_compute_cells_and_kzg_proofs(oracle, oracle_noprecomp, precomp, name, case):
    o = oracle if precomp else oracle_noprecomp
    out = _call(o.compute_cells_and_kzg_proofs, case["input"]["blob"])
    exp = case["output"]
    if exp is None:
        assert out is None
        assert _call(o.compute_cells, case["input"]["blob"]) is None
        return
    assert out is not None
    assert out[0] == exp[0] and out[1] == exp[1]
    # compute_cells == cells half (tests/compute_cells_and_kzg_proofs.rs:94-99)
    assert o.compute_cells(case["input"]["blob"]) == exp[0]
    # data_is_contained_in_the_first_section_of_cells (fk20/prover.rs:251-275)
    assert b"".join(out[0][:64]) == case["input"]["blob"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("verify_cell_kzg_proof_batch").items()))
def test_verify_cell_kzg_proof_batch(oracle, name, case):
    i = case["input"]
    out = _call(oracle.verify_cell_kzg_proof_batch, i["commitments"], i["cell_indices"], i["cells"], i["proofs"])
    assert out == case["output"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("recover_cells_and_kzg_proofs").items()))
def test_recover_cells_and_kzg_proofs(oracle, name, case):
    i = case["input"]
    out = _call(oracle.recover_cells_and_kzg_proofs, i["cell_indices"], i["cells"])
    exp = case["output"]
    if exp is None:
        assert out is None
    else:
        assert out is not None and out[0] == exp[0] and out[1] == exp[1]


# ---- EIP-4844 single-point families (crates/eip4844/tests/*.rs) ----
@pytest.mark.parametrize("name,case", sorted(vectors.load("compute_kzg_proof").items()))
def test_compute_kzg_proof(oracle, name, case):
    i = case["input"]
    out = _call(oracle.compute_kzg_proof, i["blob"], i["z"])
    assert (list(out) if out is not None else None) == case["output"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("compute_blob_kzg_proof").items()))
def test_compute_blob_kzg_proof(oracle, name, case):
    i = case["input"]
    assert _call(oracle.compute_blob_kzg_proof, i["blob"], i["commitment"]) == case["output"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("verify_kzg_proof").items()))
def test_verify_kzg_proof(oracle, name, case):
    i = case["input"]
    assert _call(oracle.verify_kzg_proof, i["commitment"], i["z"], i["y"], i["proof"]) == case["output"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("verify_blob_kzg_proof").items()))
def test_verify_blob_kzg_proof(oracle, name, case):
    i = case["input"]
    assert _call(oracle.verify_blob_kzg_proof, i["blob"], i["commitment"], i["proof"]) == case["output"]


@pytest.mark.parametrize("name,case", sorted(vectors.load("verify_blob_kzg_proof_batch").items()))
def test_verify_blob_kzg_proof_batch(oracle, name, case):
    i = case["input"]
    assert _call(oracle.verify_blob_kzg_proof_batch, i["blobs"], i["commitments"], i["proofs"]) == case["output"]
