"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/c_eth_kzg.h declares (no compute call is made here -- those need a GPU)."""
import ctypes
import importlib
import os
import re

import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kzg = importlib.import_module("rust-eth-kzg_amd")
if not os.path.exists(kzg.LIB_PATH):  # fresh checkout: cross-compile the HIP extension (no GPU needed, a few minutes)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "rust-eth-kzg_amd", "csrc"), "-j", str(min(8, os.cpu_count() or 1))])


def _declared_symbols(header="c_eth_kzg.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eth_kzg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_python_symbol_lists_agree():
    assert _declared_symbols() == sorted(kzg.EXPORTED_SYMBOLS)
    assert _declared_symbols("c_eth_kzg_test_hooks.h") == sorted(kzg.TEST_HOOK_SYMBOLS)
    assert not set(kzg.TEST_HOOK_SYMBOLS) & set(_declared_symbols()), "test hooks leaked into the drop-in header"


def test_library_exports_every_test_hook():
    lib = ctypes.CDLL(kzg.LIB_PATH)
    for name in _declared_symbols("c_eth_kzg_test_hooks.h"):
        assert hasattr(lib, name), name


def test_reference_symbols_are_all_there():
    """The 16 functions of the reference's C ABI (bindings/c/src/lib.rs:79-566), by name."""
    ref = ["eth_kzg_das_context_new", "eth_kzg_das_context_free", "eth_kzg_free_error_message", "eth_kzg_blob_to_kzg_commitment",
           "eth_kzg_compute_cells_and_kzg_proofs", "eth_kzg_compute_cells", "eth_kzg_verify_cell_kzg_proof_batch",
           "eth_kzg_recover_cells_and_proofs", "eth_kzg_constant_bytes_per_cell", "eth_kzg_constant_bytes_per_proof",
           "eth_kzg_constant_cells_per_ext_blob", "eth_kzg_compute_kzg_proof", "eth_kzg_compute_blob_kzg_proof",
           "eth_kzg_verify_kzg_proof", "eth_kzg_verify_blob_kzg_proof", "eth_kzg_verify_blob_kzg_proof_batch"]
    assert set(ref) <= set(_declared_symbols())


def test_static_archive_is_built_and_defines_the_abi():
    """The reference ships cdylib AND staticlib (bindings/c/Cargo.toml:12); its Go binding links the archive
    (bindings/golang/prover.go:4-8).  libc_eth_kzg.a must exist and define every declared symbol."""
    a = os.path.join(ROOT, "rust-eth-kzg_amd", "libc_eth_kzg.a")
    assert os.path.exists(a), "make -C rust-eth-kzg_amd/csrc builds it next to the .so"
    out = subprocess.run(["nm", "--defined-only", a], capture_output=True, text=True).stdout
    defined = set(re.findall(r" T (eth_kzg_[a-z0-9_]+)", out))
    assert set(_declared_symbols()) <= defined


def test_library_exports_every_declared_symbol():
    assert os.path.exists(kzg.LIB_PATH), "build the HIP extension first (__graft_entry__.build())"
    lib = ctypes.CDLL(kzg.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/c_eth_kzg.h but not exported"


def test_constants_without_gpu():
    lib = kzg.load_library()
    assert lib.eth_kzg_constant_bytes_per_cell() == 2048
    assert lib.eth_kzg_constant_bytes_per_proof() == 48
    assert lib.eth_kzg_constant_cells_per_ext_blob() == 128
    lib.eth_kzg_das_context_free(None)  # NULL-safe, as in the reference (bindings/c/src/lib.rs:109-116)
    lib.eth_kzg_free_error_message(None)
