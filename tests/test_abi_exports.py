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


def test_hooks_library_exports_every_test_hook_and_the_product_none():
    """the stage-level hooks live in libc_eth_kzg_hooks.so (what tests/ loads); the product library -- .so and .a --
    neither exports nor defines any eth_kzg_amd_test_* symbol, nor the test kernels behind them."""
    assert os.path.exists(kzg.HOOKS_LIB_PATH), "make -C rust-eth-kzg_amd/csrc builds it next to the product library"
    hooks = ctypes.CDLL(kzg.HOOKS_LIB_PATH)
    for name in _declared_symbols("c_eth_kzg_test_hooks.h"):
        assert hasattr(hooks, name), name
    for name in _declared_symbols():  # the hooks library is a superset: the whole drop-in ABI too
        assert hasattr(hooks, name), name
    for path in (kzg.LIB_PATH, os.path.join(ROOT, "rust-eth-kzg_amd", "libc_eth_kzg.a")):
        flags = ["--defined-only"] + (["-D"] if path.endswith(".so") else [])
        out = subprocess.run(["nm"] + flags + [path], capture_output=True, text=True).stdout
        assert "eth_kzg_amd_test_" not in out and "k_test_" not in out and "test_fixed_msm" not in out, path


def test_oversized_counts_are_rejected_not_crashed():
    """the batched entry points bound their count BEFORE anything else is looked at -- n = 2^32 (which the
    engine's `int` would have read as 0 and answered Ok with untouched outputs) is Err(InvalidInput).  No GPU needed: the check
    comes first, so a NULL context is never reached.  The contract: malformed input is Err, never a crash
    (bindings/c/src/lib.rs:272-280)."""
    lib = kzg.load_library()
    n = 1 << 32

    def expect_invalid(res):
        assert res.status == 1 and res.error_msg
        msg = ctypes.string_at(res.error_msg).decode()
        lib.eth_kzg_free_error_message(res.error_msg)
        assert msg.startswith("InvalidInput"), msg

    expect_invalid(lib.eth_kzg_amd_compute_cells_and_kzg_proofs_batch(None, n, None, None, None, None))
    expect_invalid(lib.eth_kzg_amd_blob_to_kzg_commitment_batch(None, n, None, None, None))
    expect_invalid(lib.eth_kzg_amd_recover_cells_and_proofs_batch(None, n, None, None, None, None, None, None, None))
    expect_invalid(lib.eth_kzg_amd_verify_cell_kzg_proof_batch_many(None, n, None, None, None, None, None, None, None, None, None, None))
    expect_invalid(lib.eth_kzg_amd_compute_cells_and_kzg_proofs_device(None, n, None, None, None, None, None))
    expect_invalid(lib.eth_kzg_amd_blob_to_kzg_commitment_device(None, n, None, None, None, None))
    expect_invalid(lib.eth_kzg_amd_recover_cells_and_proofs_device(None, (1 << 24) + 1, None, None, None, None, None, None))
    expect_invalid(lib.eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi(None, 1, n, None, None, None, None))
    assert lib.eth_kzg_amd_abi_version() == 6


def test_device_list_constructor_reports_a_missing_gpu():
    """eth_kzg_amd_das_context_new_on_devices never aborts: an empty list, an oversized one and a GPU that does not exist are NULL +
    a message (in a child process, so that a regression that aborts fails the test instead of the run)."""
    import sys
    code = (
        "import ctypes as C, importlib, sys\n"
        "import numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "kzg = importlib.import_module('rust-eth-kzg_amd')\n"
        "lib = kzg.load_library()\n"
        "for devs in ([], [4094, 4095], list(range(65))):\n"
        "    res = kzg.CResult()\n"
        "    a = np.array(devs, dtype=np.int32)\n"
        "    p = lib.eth_kzg_amd_das_context_new_on_devices(True, a.ctypes.data_as(C.c_void_p) if len(devs) else None, len(devs), 8.0, C.byref(res))\n"
        "    assert not p and res.status == 1 and res.error_msg, devs\n"
        "    msg = C.cast(res.error_msg, C.c_char_p).value.decode()\n"
        "    lib.eth_kzg_free_error_message(res.error_msg)\n"
        "    assert msg.startswith('ContextCreation('), msg\n"
        "try:\n"
        "    kzg.DASContext(use_precomp=True, devices=[4094, 4095])\n"
        "    raise SystemExit('no error from the Python wrapper')\n"
        "except kzg.KzgError as e:\n"
        "    assert 'ContextCreation' in str(e)\n"
        "print('device list ok')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "device list ok" in out.stdout, out.stdout + out.stderr


def _header_prototypes():
    """include/c_eth_kzg.h -> {symbol: {"ret": type, "args": [types], "arg_names": [names]}} with const dropped and the fixed-width
    names shortened (uint8_t* -> uint8*): the shape an FFI binds, which is what the reference's generated mirrors record."""
    import json  # noqa: F401
    text = open(os.path.join(ROOT, "include", "c_eth_kzg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(eth_kzg_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        def canon(t):
            t = re.sub(r"\bconst\b", "", t)
            t = re.sub(r"\s+", "", t)
            return t.replace("uint8_t", "uint8").replace("uint64_t", "uint64")
        args, names = [], []
        for a in [x.strip() for x in m.group(3).split(",") if x.strip() and x.strip() != "void"]:
            mm = re.match(r"(.*?)(\w+)$", a)
            args.append(canon(mm.group(1)))
            names.append(mm.group(2))
        protos[m.group(2)] = {"ret": canon(m.group(1)), "args": args, "arg_names": names}
    return protos


def test_reference_signatures_match_the_generated_mirrors():
    """All 16 functions of the reference's C ABI (bindings/c/src/lib.rs:79-566): return type, argument count, order, pointer depth and
    integer width of include/c_eth_kzg.h against the reference's own machine-generated mirrors -- the C# P/Invoke table
    (typed: byte*, byte**, ulong, ulong*, bool*) and the Nim header (pointer depths) -- as parsed into
    tests/golden/abi_signatures.json by tools/gen_abi_signatures.py.  `char*` and `uint8*` are the same thing to an FFI."""
    import json
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "abi_signatures.json")))
    ours = _header_prototypes()
    assert len(golden["csharp"]) == 16 and sorted(golden["csharp"]) == sorted(golden["nim"])
    for name, ref in golden["csharp"].items():
        assert name in ours, name
        mine = ours[name]
        same = lambda a, b: a == b or {a, b} == {"char*", "uint8*"}  # noqa: E731
        assert same(mine["ret"], ref["ret"]), (name, mine["ret"], ref["ret"])
        assert len(mine["args"]) == len(ref["args"]), (name, mine["args"], ref["args"])
        for k, (a, b) in enumerate(zip(mine["args"], ref["args"])):
            assert same(a, b), (name, k, a, b)
        assert [n.lstrip("@") for n in mine["arg_names"]] == ref["arg_names"], (name, mine["arg_names"], ref["arg_names"])
        nim = golden["nim"][name]
        assert len(nim["args"]) == len(mine["args"])
        for k, (a, b) in enumerate(zip(mine["args"], nim["args"])):
            depth = a.count("*")
            if b == "ptr":
                assert depth == 1, (name, k, a)
            elif b == "ptr*":
                assert depth == 2, (name, k, a)
            else:
                assert a == b, (name, k, a, b)
        assert nim["ret"] in (mine["ret"], "ptr")


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


def test_try_new_reports_a_missing_gpu_instead_of_aborting():
    """eth_kzg_amd_das_context_try_new: where eth_kzg_das_context_new aborts the process (the reference panics across the FFI,
    bindings/c/src/lib.rs:79-92), this constructor returns NULL and an error message -- on this GPU-less container for every
    ordinal, on a GPU box for an ordinal that does not exist (run in a child process so that a regression that aborts fails the
    test instead of the test run)."""
    import sys
    code = (
        "import ctypes as C, importlib, sys\n"
        "sys.path.insert(0, %r)\n"
        "kzg = importlib.import_module('rust-eth-kzg_amd')\n"
        "lib = kzg.load_library()\n"
        "res = kzg.CResult()\n"
        "p = lib.eth_kzg_amd_das_context_try_new(True, 4095, 8.0, C.byref(res))\n"
        "assert not p, 'a context on GPU ordinal 4095?'\n"
        "assert res.status == 1 and res.error_msg\n"
        "msg = C.cast(res.error_msg, C.c_char_p).value.decode()\n"
        "lib.eth_kzg_free_error_message(res.error_msg)\n"
        "assert msg.startswith('ContextCreation('), msg\n"
        "try:\n"
        "    kzg.DASContext(use_precomp=True, device=4095, table_budget_gb=8)\n"
        "    raise SystemExit('no error from the Python wrapper')\n"
        "except kzg.KzgError as e:\n"
        "    assert 'ContextCreation' in str(e)\n"
        "print('try_new ok:', msg)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "try_new ok" in out.stdout, out.stdout + out.stderr
