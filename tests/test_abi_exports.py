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


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "c_eth_kzg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eth_kzg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_python_symbol_lists_agree():
    assert _declared_symbols() == sorted(kzg.EXPORTED_SYMBOLS)


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
