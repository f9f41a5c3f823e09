"""rust-eth-kzg_amd -- Python binding of the MI355X-native KZG engine (libc_eth_kzg.so).

Host-side mirror of the reference's `DASContext` API (reference: crates/eip7594/src/lib.rs:41-87,
prover.rs:100-171, verifier.rs:72-112) over the drop-in C ABI declared in include/c_eth_kzg.h.
Same method names, argument meaning and error behaviour as the reference bindings: an operation
that the reference reports as `Err` raises `KzgError`; `verify_cell_kzg_proof_batch` returns False
for an invalid proof and raises for malformed input.

There is NO CPU fallback: importing works without a GPU, but creating a context requires the HIP
library built by `rust-eth-kzg_amd/csrc/Makefile` and a gfx950 device, and fails loudly otherwise.

(The directory name contains a hyphen, so load it with importlib:
    kzg = importlib.import_module("rust-eth-kzg_amd") )
"""
import ctypes as C

import numpy as np
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libc_eth_kzg.so")  # the product library: what a maintainer links
HOOKS_LIB_PATH = os.path.join(_HERE, "libc_eth_kzg_hooks.so")  # the same objects + the stage-level test hooks: what tests/ loads

BYTES_PER_BLOB = 131072
BYTES_PER_CELL = 2048
BYTES_PER_COMMITMENT = 48
BYTES_PER_PROOF = 48
CELLS_PER_EXT_BLOB = 128
FIELD_ELEMENTS_PER_BLOB = 4096
FIELD_ELEMENTS_PER_CELL = 64


class KzgError(Exception):
    """An operation the reference would report as `Err(..)` (CResult.status == Err)."""


class CResult(C.Structure):
    _fields_ = [("status", C.c_int), ("error_msg", C.c_void_p)]


# every symbol include/c_eth_kzg.h declares (tests assert the library exports all of them)
VERIFY_PARTIAL_BYTES = 96

EXPORTED_SYMBOLS = [
    "eth_kzg_das_context_new", "eth_kzg_das_context_free", "eth_kzg_free_error_message",
    "eth_kzg_blob_to_kzg_commitment", "eth_kzg_compute_cells_and_kzg_proofs", "eth_kzg_compute_cells",
    "eth_kzg_verify_cell_kzg_proof_batch", "eth_kzg_recover_cells_and_proofs",
    "eth_kzg_constant_bytes_per_cell", "eth_kzg_constant_bytes_per_proof", "eth_kzg_constant_cells_per_ext_blob",
    "eth_kzg_compute_kzg_proof", "eth_kzg_compute_blob_kzg_proof", "eth_kzg_verify_kzg_proof",
    "eth_kzg_verify_blob_kzg_proof", "eth_kzg_verify_blob_kzg_proof_batch",
    "eth_kzg_amd_das_context_new_on_device",
    "eth_kzg_amd_das_context_try_new", "eth_kzg_amd_das_context_new_on_devices", "eth_kzg_amd_context_devices",
    "eth_kzg_amd_abi_version", "eth_kzg_amd_window_count", "eth_kzg_amd_glv_table",
    "eth_kzg_amd_compute_cells_and_kzg_proofs_batch", "eth_kzg_amd_blob_to_kzg_commitment_batch",
    "eth_kzg_amd_recover_cells_and_proofs_batch", "eth_kzg_amd_recover_cells_and_proofs_device",
    "eth_kzg_amd_verify_cell_kzg_proof_batch_partial", "eth_kzg_amd_verify_cell_kzg_proof_batch_combine",
    "eth_kzg_amd_compute_cells_and_kzg_proofs_device", "eth_kzg_amd_blob_to_kzg_commitment_device",
    "eth_kzg_amd_verify_cell_kzg_proof_batch_device", "eth_kzg_amd_verify_cell_kzg_proof_batch_many",
    "eth_kzg_amd_table_bytes", "eth_kzg_amd_window_bits", "eth_kzg_amd_tables_ready", "eth_kzg_amd_table_groups_ready", "eth_kzg_amd_table_build_info", "eth_kzg_amd_linmap_info",
    "eth_kzg_amd_set_profiling", "eth_kzg_amd_get_stage_times",
    "eth_kzg_amd_comm_probe", "eth_kzg_amd_comm_unique_id", "eth_kzg_amd_comm_init", "eth_kzg_amd_comm_info",
    "eth_kzg_amd_all_gather", "eth_kzg_amd_comm_destroy",
    "eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi", "eth_kzg_amd_device_count",
]
# include/c_eth_kzg_test_hooks.h: stage-level hooks for tests/, not part of the drop-in ABI
TEST_HOOK_SYMBOLS = [
    "eth_kzg_amd_test_fr_ntt4096", "eth_kzg_amd_test_g1_fft128", "eth_kzg_amd_test_fixed_msm",
    "eth_kzg_amd_test_g1_decompress", "eth_kzg_amd_test_field_mul",
]

_lib = None


def load_library():
    """dlopen libc_eth_kzg.so and declare the prototypes. Raises if the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ETH_KZG_AMD_LIB", LIB_PATH)  # A/B builds of the same library (tools/); default: the in-tree build
    if not os.path.exists(path):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C rust-eth-kzg_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(path)
    P, U8P, U64 = C.c_void_p, C.c_char_p, C.c_uint64
    lib.eth_kzg_das_context_new.restype = P
    lib.eth_kzg_das_context_new.argtypes = [C.c_bool]
    lib.eth_kzg_amd_das_context_new_on_device.restype = P
    lib.eth_kzg_amd_das_context_new_on_device.argtypes = [C.c_bool, C.c_int]
    lib.eth_kzg_amd_das_context_try_new.restype = P
    lib.eth_kzg_amd_das_context_try_new.argtypes = [C.c_bool, C.c_int, C.c_double, C.POINTER(CResult)]
    lib.eth_kzg_das_context_free.argtypes = [P]
    lib.eth_kzg_free_error_message.argtypes = [P]
    for name, args in {
        "eth_kzg_blob_to_kzg_commitment": [P, U8P, P],
        "eth_kzg_compute_cells_and_kzg_proofs": [P, U8P, P, P],
        "eth_kzg_compute_cells": [P, U8P, P],
        "eth_kzg_verify_cell_kzg_proof_batch": [P, U64, P, U64, P, U64, P, U64, P, P],
        "eth_kzg_recover_cells_and_proofs": [P, U64, P, U64, P, P, P],
        "eth_kzg_compute_kzg_proof": [P, U8P, U8P, P, P],
        "eth_kzg_compute_blob_kzg_proof": [P, U8P, U8P, P],
        "eth_kzg_verify_kzg_proof": [P, U8P, U8P, U8P, U8P, P],
        "eth_kzg_verify_blob_kzg_proof": [P, U8P, U8P, U8P, P],
        "eth_kzg_verify_blob_kzg_proof_batch": [P, U64, P, U64, P, U64, P, P],
        "eth_kzg_amd_compute_cells_and_kzg_proofs_batch": [P, U64, P, P, P, P],
        "eth_kzg_amd_blob_to_kzg_commitment_batch": [P, U64, P, P, P],
        "eth_kzg_amd_recover_cells_and_proofs_batch": [P, U64, P, P, P, P, P, P, P],
        "eth_kzg_amd_verify_cell_kzg_proof_batch_partial": [P, U64, P, U64, P, U64, P, U64, P, U64, U64, P],
        "eth_kzg_amd_verify_cell_kzg_proof_batch_combine": [P, U64, U8P, P],
        "eth_kzg_amd_recover_cells_and_proofs_device": [P, U64, P, P, P, P, P, P],
        "eth_kzg_amd_compute_cells_and_kzg_proofs_device": [P, U64, P, P, P, P, P],
        "eth_kzg_amd_blob_to_kzg_commitment_device": [P, U64, P, P, P, P],
        "eth_kzg_amd_verify_cell_kzg_proof_batch_device": [P, U64, P, P, P, P, P, P],
        "eth_kzg_amd_verify_cell_kzg_proof_batch_many": [P, U64, P, P, P, P, P, P, P, P, P, P],
        "eth_kzg_amd_comm_unique_id": [P],
        "eth_kzg_amd_comm_probe": [P, P, U64],
        "eth_kzg_amd_comm_info": [P, P, P],
        "eth_kzg_amd_comm_init": [P, U8P, C.c_int, C.c_int],
        "eth_kzg_amd_all_gather": [P, P, P, U64, P],
        "eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi": [P, U64, U64, P, P, P, P],
    }.items():
        fn = getattr(lib, name)
        fn.restype = CResult
        fn.argtypes = args
    for name in ("eth_kzg_constant_bytes_per_cell", "eth_kzg_constant_bytes_per_proof",
                 "eth_kzg_constant_cells_per_ext_blob"):
        getattr(lib, name).restype = U64
    lib.eth_kzg_amd_comm_destroy.argtypes = [P]
    lib.eth_kzg_amd_comm_destroy.restype = None
    lib.eth_kzg_amd_device_count.restype = C.c_int
    lib.eth_kzg_amd_table_bytes.restype = U64
    lib.eth_kzg_amd_table_bytes.argtypes = [P]
    lib.eth_kzg_amd_window_bits.argtypes = [P]
    lib.eth_kzg_amd_tables_ready.argtypes = [P, C.c_int]
    lib.eth_kzg_amd_tables_ready.restype = C.c_int
    lib.eth_kzg_amd_table_build_info.argtypes = [P, C.POINTER(C.c_double)]
    lib.eth_kzg_amd_table_build_info.restype = None
    lib.eth_kzg_amd_table_groups_ready.argtypes = [P]
    lib.eth_kzg_amd_table_groups_ready.restype = C.c_int
    lib.eth_kzg_amd_linmap_info.argtypes = [P, P]
    lib.eth_kzg_amd_linmap_info.restype = None
    lib.eth_kzg_amd_set_profiling.argtypes = [P, C.c_int]
    lib.eth_kzg_amd_set_profiling.restype = None
    lib.eth_kzg_amd_get_stage_times.argtypes = [P, P, P, C.c_int]
    lib.eth_kzg_amd_das_context_new_on_devices.restype = P
    lib.eth_kzg_amd_das_context_new_on_devices.argtypes = [C.c_bool, P, U64, C.c_double, C.POINTER(CResult)]
    lib.eth_kzg_amd_context_devices.restype = U64
    lib.eth_kzg_amd_context_devices.argtypes = [P, P, U64]
    lib.eth_kzg_amd_abi_version.restype = C.c_int
    lib.eth_kzg_amd_window_count.argtypes = [P]
    lib.eth_kzg_amd_glv_table.argtypes = [P]
    # the stage-level test hooks exist in libc_eth_kzg_hooks.so only (what tests/ loads through ETH_KZG_AMD_LIB); the product
    # library does not export them
    for name, args in {
        "eth_kzg_amd_test_fr_ntt4096": [P, U8P, P, C.c_int],
        "eth_kzg_amd_test_g1_fft128": [P, U8P, P, C.c_int, C.c_int],
        "eth_kzg_amd_test_fixed_msm": [P, U8P, C.c_int, P],
        "eth_kzg_amd_test_g1_decompress": [P, U8P, C.c_int, C.c_int, P, P],
        "eth_kzg_amd_test_field_mul": [P, U8P, U8P, P, C.c_int, C.c_int],
    }.items():
        if hasattr(lib, name):
            getattr(lib, name).argtypes = args
    _lib = lib
    return lib


def present_masks(n, present):
    """Two 64-bit words per blob for eth_kzg_amd_recover_cells_and_proofs_device: bit (c % 64) of word (c / 64) set when
    cell c of that blob carries data.  present[b] = iterable of cell indices (lists, tuples, rows of a 2-D array ...)."""
    masks = np.zeros(2 * max(1, n), dtype=np.uint64)
    # the same index list for many blobs (the usual case: [idx] * n) is folded into its mask once.  The cache holds the
    # row OBJECT and compares with `is`: an id() of a temporary (a row of a 2-D numpy array is a fresh view per access)
    # can be reused by the next row once the temporary is freed, which would silently give every blob the first mask.
    prev_row, prev_mask = None, 0
    for b in range(n):
        row = present[b]
        if row is prev_row and prev_row is not None:
            m = prev_mask
        else:
            m = 0
            for c in row:
                c = int(c)
                if not 0 <= c < CELLS_PER_EXT_BLOB:
                    raise KzgError("InvalidCellIndex")
                m |= 1 << c
            prev_row, prev_mask = row, m
        masks[2 * b] = m & 0xFFFFFFFFFFFFFFFF
        masks[2 * b + 1] = m >> 64
    return masks


def _ptr_array(bufs):
    """(array of char*, keep-alive list) over a list of bytes / ctypes buffers."""
    arr = (C.c_void_p * max(1, len(bufs)))()
    keep = []
    for i, b in enumerate(bufs):
        if isinstance(b, (bytes, bytearray)):
            b = C.create_string_buffer(bytes(b), len(b))
        keep.append(b)
        arr[i] = C.addressof(b)
    return arr, keep


def _flat_ptrs(items, size):
    """(uint64 pointer table, keep-alive) over ONE contiguous copy of equal-length byte strings: marshalling thousands
    of cells through per-item ctypes buffers costs more than the GPU work it feeds."""
    n = len(items)
    flat = np.frombuffer(b"".join(bytes(x) if not isinstance(x, bytes) else x for x in items), dtype=np.uint8) if n else np.zeros(1, np.uint8)
    ptrs = np.uint64(flat.ctypes.data) + np.arange(max(1, n), dtype=np.uint64) * np.uint64(size)
    return ptrs, flat


def _out_ptrs(n_outer, n_inner, size):
    """Caller-allocated outputs as one flat buffer: (flat uint8 array, inner pointer tables [n_outer][n_inner],
    outer pointer table [n_outer]) in the layout the C ABI takes (arrays of n_inner pointers per blob)."""
    flat = np.zeros(max(1, n_outer * n_inner * size), dtype=np.uint8)
    inner = np.uint64(flat.ctypes.data) + np.arange(max(1, n_outer * n_inner), dtype=np.uint64) * np.uint64(size)
    outer = np.uint64(inner.ctypes.data) + np.arange(max(1, n_outer), dtype=np.uint64) * np.uint64(n_inner * 8)
    return flat, inner, outer


def _split(flat, n_outer, n_inner, size):
    raw = flat.tobytes()
    return [[raw[(b * n_inner + k) * size:(b * n_inner + k + 1) * size] for k in range(n_inner)] for b in range(n_outer)]


def _vp(arr):
    return arr.ctypes.data_as(C.c_void_p)


class DASContext:
    """Mirror of `rust_eth_kzg::DASContext` (crates/eip7594/src/lib.rs:41-87)."""

    device_index = 0

    def __init__(self, use_precomp=True, device=None, wait_tables=True, table_budget_gb=None, devices=None):
        """devices: a device LIST -- one engine per listed GPU behind this one context (eth_kzg_amd_das_context_new_on_devices:
        single calls go to the least-loaded device, batched calls are cut into contiguous slices); ordinals may repeat.
        table_budget_gb: HBM for the two window tables together (None: $ETH_KZG_AMD_TABLE_GB or the library's default of 108 GB;
        a negative number: whatever the HBM holds) -- given, the context is made by eth_kzg_amd_das_context_try_new, which reports a
        failure as KzgError instead of aborting the process.
        wait_tables: the C entry point returns as soon as the start tables are up (progressive start); by default this
        wrapper then waits for the wide tables, so that timing and table introspection see the final state.  Pass False to
        use the context at once (results are identical on every table)."""
        self.device_index = int(device) if device is not None else int(os.environ.get("ETH_KZG_AMD_DEVICE", "0"))
        self._lib = load_library()
        if devices is not None:
            devs = np.array([int(d) for d in devices], dtype=np.int32)
            self.device_index = int(devs[0]) if len(devs) else 0
            res = CResult()
            self._ctx = C.c_void_p(self._lib.eth_kzg_amd_das_context_new_on_devices(
                bool(use_precomp), _vp(devs) if len(devs) else None, len(devs), float(table_budget_gb or 0.0), C.byref(res)))
            if not self._ctx.value:
                msg = C.cast(res.error_msg, C.c_char_p).value.decode() if res.error_msg else "unknown"
                self._lib.eth_kzg_free_error_message(res.error_msg)
                raise KzgError(msg)
        elif table_budget_gb is not None:
            res = CResult()
            self._ctx = C.c_void_p(self._lib.eth_kzg_amd_das_context_try_new(bool(use_precomp), self.device_index, float(table_budget_gb), C.byref(res)))
            if not self._ctx.value:
                msg = C.cast(res.error_msg, C.c_char_p).value.decode() if res.error_msg else "unknown"
                self._lib.eth_kzg_free_error_message(res.error_msg)
                raise KzgError(msg)
        elif device is None:
            self._ctx = C.c_void_p(self._lib.eth_kzg_das_context_new(bool(use_precomp)))
        else:
            self._ctx = C.c_void_p(self._lib.eth_kzg_amd_das_context_new_on_device(bool(use_precomp), int(device)))
        if not self._ctx.value:
            raise RuntimeError("eth_kzg_das_context_new returned NULL")
        if wait_tables:
            self.tables_ready(-1)

    def devices(self):
        """The context's device list (eth_kzg_amd_context_devices)."""
        out = np.zeros(64, dtype=np.int32)
        n = int(self._lib.eth_kzg_amd_context_devices(self._ctx, _vp(out), 64))
        return [int(d) for d in out[:n]]

    def tables_ready(self, wait_ms=0):
        """1 = final window tables in use, 0 = still on the start tables, 2 = the wide build failed (stays on what it has)."""
        return int(self._lib.eth_kzg_amd_tables_ready(self._ctx, int(wait_ms)))

    def table_build_info(self):
        """{'hipmalloc_ms', 'longest_hipmalloc_ms', 'pieces', 'bytes'} of the window tables in use (eth_kzg_amd_table_build_info)."""
        out = (C.c_double * 4)()
        self._lib.eth_kzg_amd_table_build_info(self._ctx, out)
        return {"hipmalloc_ms": round(out[0], 1), "longest_hipmalloc_ms": round(out[1], 1), "pieces": int(out[2]), "bytes": int(out[3])}

    def table_groups_ready(self):
        """How many of the 128 MSM groups already run on the wide FK20 table under construction (128: complete / nothing being built)."""
        return int(self._lib.eth_kzg_amd_table_groups_ready(self._ctx))

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.eth_kzg_das_context_free(self._ctx)
            self._ctx = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._ctx

    def _check(self, res):
        if res.status != 0:
            msg = C.string_at(res.error_msg).decode() if res.error_msg else "error"
            self._lib.eth_kzg_free_error_message(res.error_msg)
            raise KzgError(msg)

    # ---- reference API -------------------------------------------------------------------
    def blob_to_kzg_commitment(self, blob):
        if len(blob) != BYTES_PER_BLOB:  # the reference bindings reject a wrong-length slice before the FFI call
            raise KzgError("BlobHasInvalidLength")
        out = C.create_string_buffer(48)
        self._check(self._lib.eth_kzg_blob_to_kzg_commitment(self._ctx, bytes(blob), out))
        return out.raw

    def compute_cells_and_kzg_proofs(self, blob):
        if len(blob) != BYTES_PER_BLOB:
            raise KzgError("BlobHasInvalidLength")
        cells = [C.create_string_buffer(BYTES_PER_CELL) for _ in range(CELLS_PER_EXT_BLOB)]
        proofs = [C.create_string_buffer(48) for _ in range(CELLS_PER_EXT_BLOB)]
        ca, _k1 = _ptr_array(cells)
        pa, _k2 = _ptr_array(proofs)
        self._check(self._lib.eth_kzg_compute_cells_and_kzg_proofs(self._ctx, bytes(blob), ca, pa))
        return [c.raw for c in cells], [p.raw for p in proofs]

    def compute_cells(self, blob):
        if len(blob) != BYTES_PER_BLOB:
            raise KzgError("BlobHasInvalidLength")
        cells = [C.create_string_buffer(BYTES_PER_CELL) for _ in range(CELLS_PER_EXT_BLOB)]
        ca, _k = _ptr_array(cells)
        self._check(self._lib.eth_kzg_compute_cells(self._ctx, bytes(blob), ca))
        return [c.raw for c in cells]

    def verify_cell_kzg_proof_batch(self, commitments, cell_indices, cells, proofs):
        if any(len(c) != 48 for c in commitments) or any(len(p) != 48 for p in proofs) \
                or any(len(c) != BYTES_PER_CELL for c in cells):
            raise KzgError("InvalidLength")
        ca, _k1 = _flat_ptrs(commitments, 48)
        cla, _k2 = _flat_ptrs(cells, BYTES_PER_CELL)
        pa, _k3 = _flat_ptrs(proofs, 48)
        idx = np.array(cell_indices, dtype=np.uint64) if len(cell_indices) else np.zeros(1, np.uint64)
        ok = C.c_bool(False)
        self._check(self._lib.eth_kzg_verify_cell_kzg_proof_batch(
            self._ctx, len(commitments), _vp(ca), len(cell_indices), _vp(idx), len(cells), _vp(cla), len(proofs), _vp(pa),
            C.byref(ok)))
        return bool(ok.value)

    def prepare_verify_cell_kzg_proof_batch(self, commitments, cell_indices, cells, proofs):
        """Marshal a verification batch into the C ABI's pointer tables ONCE and return a zero-argument callable that runs
        eth_kzg_verify_cell_kzg_proof_batch on them: what a C caller's loop costs, without this wrapper's copying."""
        if any(len(c) != 48 for c in commitments) or any(len(p) != 48 for p in proofs) \
                or any(len(c) != BYTES_PER_CELL for c in cells):
            raise KzgError("InvalidLength")
        ca, k1 = _flat_ptrs(commitments, 48)
        cla, k2 = _flat_ptrs(cells, BYTES_PER_CELL)
        pa, k3 = _flat_ptrs(proofs, 48)
        idx = np.array(cell_indices, dtype=np.uint64) if len(cell_indices) else np.zeros(1, np.uint64)
        n = (len(commitments), len(cell_indices), len(cells), len(proofs))

        def run(_keep=(k1, k2, k3)):
            ok = C.c_bool(False)
            self._check(self._lib.eth_kzg_verify_cell_kzg_proof_batch(
                self._ctx, n[0], _vp(ca), n[1], _vp(idx), n[2], _vp(cla), n[3], _vp(pa), C.byref(ok)))
            return bool(ok.value)
        return run

    def prepare_verify_cell_kzg_proof_batch_many(self, problems):
        """problems = [(commitments, cell_indices, cells, proofs), ...]: marshal them into the pointer tables of
        eth_kzg_amd_verify_cell_kzg_proof_batch_many ONCE and return a zero-argument callable that runs the call and
        returns (verified list[bool], status list[int]) -- status 0 ok, 1 bad field element, 2 bad G1 point, 3 invalid
        lengths / indices: what the single form raises as KzgError."""
        nb = len(problems)
        keep, lens = [], np.zeros((4, max(1, nb)), dtype=np.uint64)
        tabs = [np.zeros(max(1, nb), dtype=np.uint64) for _ in range(4)]  # per problem: commitments**, indices*, cells**, proofs**
        for b, (commitments, cell_indices, cells, proofs) in enumerate(problems):
            if any(len(c) != 48 for c in commitments) or any(len(p) != 48 for p in proofs) \
                    or any(len(c) != BYTES_PER_CELL for c in cells):
                # a wrong byte length cannot cross the C ABI (fixed-size buffers carry no length): the single form raises
                # InvalidLength before the call; here the problem goes in with inconsistent lengths and comes back status 3
                commitments, cell_indices, cells, proofs = [], [0], [], []
            ca, k1 = _flat_ptrs(commitments, 48)
            cla, k2 = _flat_ptrs(cells, BYTES_PER_CELL)
            pa, k3 = _flat_ptrs(proofs, 48)
            idx = np.array(cell_indices, dtype=np.uint64) if len(cell_indices) else np.zeros(1, np.uint64)
            keep += [ca, k1, cla, k2, pa, k3, idx]
            lens[:, b] = (len(commitments), len(cell_indices), len(cells), len(proofs))
            tabs[0][b], tabs[1][b], tabs[2][b], tabs[3][b] = ca.ctypes.data, idx.ctypes.data, cla.ctypes.data, pa.ctypes.data
        lens = np.ascontiguousarray(lens)
        ver = (C.c_bool * max(1, nb))()
        st = (C.c_int32 * max(1, nb))()

        def run(_keep=keep):
            self._check(self._lib.eth_kzg_amd_verify_cell_kzg_proof_batch_many(
                self._ctx, nb, _vp(lens[0]), _vp(tabs[0]), _vp(lens[1]), _vp(tabs[1]), _vp(lens[2]), _vp(tabs[2]), _vp(lens[3]), _vp(tabs[3]),
                ver, st))
            return [bool(v) for v in ver][:nb], list(st)[:nb]
        return run

    def verify_cell_kzg_proof_batch_many(self, problems):
        """Many independent verify_cell_kzg_proof_batch problems in one call (see prepare_verify_cell_kzg_proof_batch_many)."""
        return self.prepare_verify_cell_kzg_proof_batch_many(problems)()

    def verify_cell_kzg_proof_batch_partial(self, commitments, cell_indices, cells, proofs, shard_begin, shard_end):
        """This rank's share of a sharded verification: the whole batch goes in (the challenge hashes all of it), the
        cells [shard_begin, shard_end) are evaluated, 96 bytes come back (sharding.verify_cell_kzg_proof_batch_sharded)."""
        if any(len(c) != 48 for c in commitments) or any(len(p) != 48 for p in proofs) \
                or any(len(c) != BYTES_PER_CELL for c in cells):
            raise KzgError("InvalidLength")
        ca, _k1 = _flat_ptrs(commitments, 48)
        cla, _k2 = _flat_ptrs(cells, BYTES_PER_CELL)
        pa, _k3 = _flat_ptrs(proofs, 48)
        idx = np.array(cell_indices, dtype=np.uint64) if len(cell_indices) else np.zeros(1, np.uint64)
        out = C.create_string_buffer(VERIFY_PARTIAL_BYTES)
        self._check(self._lib.eth_kzg_amd_verify_cell_kzg_proof_batch_partial(
            self._ctx, len(commitments), _vp(ca), len(cell_indices), _vp(idx), len(cells), _vp(cla), len(proofs), _vp(pa),
            int(shard_begin), int(shard_end), out))
        return out.raw

    def verify_cell_kzg_proof_batch_combine(self, partials):
        """Add the gathered 96-byte partials of all ranks and run the pairing check."""
        blob = b"".join(partials)
        if len(blob) % VERIFY_PARTIAL_BYTES:
            raise KzgError("InvalidLength")
        ok = C.c_bool(False)
        self._check(self._lib.eth_kzg_amd_verify_cell_kzg_proof_batch_combine(
            self._ctx, len(blob) // VERIFY_PARTIAL_BYTES, blob, C.byref(ok)))
        return bool(ok.value)

    def recover_cells_and_kzg_proofs(self, cell_indices, cells):
        if any(len(c) != BYTES_PER_CELL for c in cells):
            raise KzgError("InvalidLength")
        cla, _k = _ptr_array(cells)
        idx = (C.c_uint64 * max(1, len(cell_indices)))(*cell_indices)
        out_cells = [C.create_string_buffer(BYTES_PER_CELL) for _ in range(CELLS_PER_EXT_BLOB)]
        out_proofs = [C.create_string_buffer(48) for _ in range(CELLS_PER_EXT_BLOB)]
        oca, _k1 = _ptr_array(out_cells)
        opa, _k2 = _ptr_array(out_proofs)
        self._check(self._lib.eth_kzg_recover_cells_and_proofs(
            self._ctx, len(cells), cla, len(cell_indices), idx, oca, opa))
        return [c.raw for c in out_cells], [p.raw for p in out_proofs]

    # ---- EIP-4844 single-point operations (crates/eip4844/src/{prover,verifier}.rs) ----
    def compute_kzg_proof(self, blob, z):
        if len(blob) != BYTES_PER_BLOB or len(z) != 32:
            raise KzgError("InvalidLength")
        proof, y = C.create_string_buffer(48), C.create_string_buffer(32)
        self._check(self._lib.eth_kzg_compute_kzg_proof(self._ctx, bytes(blob), bytes(z), proof, y))
        return proof.raw, y.raw

    def compute_blob_kzg_proof(self, blob, commitment):
        if len(blob) != BYTES_PER_BLOB or len(commitment) != 48:
            raise KzgError("InvalidLength")
        proof = C.create_string_buffer(48)
        self._check(self._lib.eth_kzg_compute_blob_kzg_proof(self._ctx, bytes(blob), bytes(commitment), proof))
        return proof.raw

    def verify_kzg_proof(self, commitment, z, y, proof):
        if len(commitment) != 48 or len(proof) != 48 or len(z) != 32 or len(y) != 32:
            raise KzgError("InvalidLength")
        ok = C.c_bool(False)
        self._check(self._lib.eth_kzg_verify_kzg_proof(self._ctx, bytes(commitment), bytes(z), bytes(y), bytes(proof), C.byref(ok)))
        return bool(ok.value)

    def verify_blob_kzg_proof(self, blob, commitment, proof):
        if len(blob) != BYTES_PER_BLOB or len(commitment) != 48 or len(proof) != 48:
            raise KzgError("InvalidLength")
        ok = C.c_bool(False)
        self._check(self._lib.eth_kzg_verify_blob_kzg_proof(self._ctx, bytes(blob), bytes(commitment), bytes(proof), C.byref(ok)))
        return bool(ok.value)

    def verify_blob_kzg_proof_batch(self, blobs, commitments, proofs):
        if any(len(b) != BYTES_PER_BLOB for b in blobs) or any(len(c) != 48 for c in commitments) \
                or any(len(p) != 48 for p in proofs):
            raise KzgError("InvalidLength")
        ba, _k1 = _ptr_array(blobs)
        ca, _k2 = _ptr_array(commitments)
        pa, _k3 = _ptr_array(proofs)
        ok = C.c_bool(False)
        self._check(self._lib.eth_kzg_verify_blob_kzg_proof_batch(self._ctx, len(blobs), ba, len(commitments), ca, len(proofs), pa,
                                                                  C.byref(ok)))
        return bool(ok.value)

    # ---- batched additions ----------------------------------------------------------------
    def compute_cells_and_kzg_proofs_batch(self, blobs):
        """List of blobs -> (status list, cells[b][128], proofs[b][128]); host buffers."""
        n = len(blobs)
        if any(len(b) != BYTES_PER_BLOB for b in blobs):
            raise KzgError("InvalidLength")
        ba, _kb = _flat_ptrs(blobs, BYTES_PER_BLOB)
        cells, _ci, cpp = _out_ptrs(n, 128, BYTES_PER_CELL)
        proofs, _pi, ppp = _out_ptrs(n, 128, 48)
        st = (C.c_int32 * max(1, n))()
        self._check(self._lib.eth_kzg_amd_compute_cells_and_kzg_proofs_batch(self._ctx, n, _vp(ba), _vp(cpp), _vp(ppp), st))
        return list(st)[:n], _split(cells, n, 128, BYTES_PER_CELL), _split(proofs, n, 128, 48)

    def host_batch_buffers(self, n):
        """Caller-side buffers of the host-pointer batch ABI, allocated and touched once so that repeated calls measure
        the library and not the page faults of fresh output memory: (cells [n,128,2048] u8, proofs [n,128,48] u8,
        cell pointer tables, proof pointer tables)."""
        cells, ci, cpp = _out_ptrs(n, 128, BYTES_PER_CELL)
        proofs, pi, ppp = _out_ptrs(n, 128, 48)
        return {"cells": cells.reshape(n, 128, BYTES_PER_CELL), "proofs": proofs.reshape(n, 128, 48), "_ci": ci, "_pi": pi,
                "cpp": cpp, "ppp": ppp, "status": (C.c_int32 * max(1, n))()}

    def compute_cells_and_kzg_proofs_batch_np(self, blobs_np, bufs, want_proofs=True):
        """eth_kzg_amd_compute_cells_and_kzg_proofs_batch on a contiguous [n,131072] uint8 array (one pointer per blob, as
        a C caller would pass) into `bufs` from host_batch_buffers; returns the status list.  Only the C call happens here."""
        n = blobs_np.shape[0]
        assert blobs_np.dtype == np.uint8 and blobs_np.flags["C_CONTIGUOUS"] and blobs_np.size == n * BYTES_PER_BLOB
        ba = np.uint64(blobs_np.ctypes.data) + np.arange(max(1, n), dtype=np.uint64) * np.uint64(BYTES_PER_BLOB)
        self._check(self._lib.eth_kzg_amd_compute_cells_and_kzg_proofs_batch(
            self._ctx, n, _vp(ba), _vp(bufs["cpp"]), _vp(bufs["ppp"]) if want_proofs else None, bufs["status"]))
        return list(bufs["status"])[:n]

    def blob_to_kzg_commitment_batch(self, blobs):
        n = len(blobs)
        if any(len(b) != BYTES_PER_BLOB for b in blobs):
            raise KzgError("InvalidLength")
        ba, _kb = _ptr_array(blobs)
        outs = [C.create_string_buffer(48) for _ in range(n)]
        oa, _ko = _ptr_array(outs)
        st = (C.c_int32 * max(1, n))()
        self._check(self._lib.eth_kzg_amd_blob_to_kzg_commitment_batch(self._ctx, n, ba, oa, st))
        return list(st)[:n], [o.raw for o in outs]

    def recover_cells_and_kzg_proofs_batch(self, batch):
        """batch = [(cell_indices, cells), ...] -> (status list, cells[b][128], proofs[b][128])."""
        n = len(batch)
        keep = []
        lens = np.array([len(c) for _, c in batch] or [0], dtype=np.uint64)
        ilens = np.array([len(i) for i, _ in batch] or [0], dtype=np.uint64)
        cpp, ipp = np.zeros(max(1, n), np.uint64), np.zeros(max(1, n), np.uint64)
        for b, (idx, cells) in enumerate(batch):
            if any(len(c) != BYTES_PER_CELL for c in cells):
                raise KzgError("InvalidLength")
            ca, k1 = _flat_ptrs(cells, BYTES_PER_CELL)
            ia = np.array(idx, dtype=np.uint64) if len(idx) else np.zeros(1, np.uint64)
            keep += [ca, k1, ia]
            cpp[b], ipp[b] = ca.ctypes.data, ia.ctypes.data
        out_cells, _ci, ocp = _out_ptrs(n, 128, BYTES_PER_CELL)
        out_proofs, _pi, opp = _out_ptrs(n, 128, 48)
        st = (C.c_int32 * max(1, n))()
        self._check(self._lib.eth_kzg_amd_recover_cells_and_proofs_batch(
            self._ctx, n, _vp(lens), _vp(cpp), _vp(ilens), _vp(ipp), _vp(ocp), _vp(opp), st))
        return list(st)[:n], _split(out_cells, n, 128, BYTES_PER_CELL), _split(out_proofs, n, 128, 48)

    def compute_cells_and_kzg_proofs_device(self, n, d_blobs, d_cells, d_proofs, want_status=True, stream=None):
        """Device-resident flat buffers (integer device addresses, e.g. torch tensor .data_ptr())."""
        st = (C.c_int32 * max(1, n))() if want_status else None
        self._check(self._lib.eth_kzg_amd_compute_cells_and_kzg_proofs_device(
            self._ctx, n, C.c_void_p(d_blobs), C.c_void_p(d_cells) if d_cells else None,
            C.c_void_p(d_proofs) if d_proofs else None, st, C.c_void_p(stream) if stream else None))
        return list(st)[:n] if want_status else None

    def recover_cells_and_kzg_proofs_device(self, n, d_cells, present, d_out_cells, d_out_proofs, stream=None):
        """Device-resident recovery: d_cells = flat [n][128][2048] buffer in HBM (integer address), present[b] = iterable
        of the cell indices of blob b that hold data.  Returns the per-blob status list; outputs stay on the device."""
        masks = present_masks(n, present)
        st = (C.c_int32 * max(1, n))()
        self._check(self._lib.eth_kzg_amd_recover_cells_and_proofs_device(
            self._ctx, n, C.c_void_p(d_cells), _vp(masks), C.c_void_p(d_out_cells) if d_out_cells else None,
            C.c_void_p(d_out_proofs) if d_out_proofs else None, st, C.c_void_p(stream) if stream else None))
        return list(st)[:n]

    # ---- multi-GPU (include/c_eth_kzg.h): the library's own RCCL communicator and the single-process fan-out
    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId (rank 0 calls this and distributes them)."""
        out = C.create_string_buffer(128)
        res = load_library().eth_kzg_amd_comm_unique_id(out)
        if res.status != 0:
            msg = C.cast(res.error_msg, C.c_char_p).value.decode() if res.error_msg else "error"
            load_library().eth_kzg_free_error_message(res.error_msg)
            raise KzgError(msg)
        return out.raw

    def comm_probe(self):
        """Local and cheap: raises KzgError unless RCCL can be bound and this context has no communicator yet; returns the
        library file that was bound (and whether it was the copy the process had already mapped)."""
        buf = C.create_string_buffer(512)
        self._check(self._lib.eth_kzg_amd_comm_probe(self._ctx, buf, 512))
        return buf.value.decode()

    def comm_init(self, unique_id, rank, world):
        """COLLECTIVE (ncclCommInitRank): every rank enters or none -- agree on comm_probe() first."""
        self._check(self._lib.eth_kzg_amd_comm_init(self._ctx, unique_id, int(rank), int(world)))

    def comm_info(self):
        """(rank, world) as the library's communicator itself reports them."""
        r, w = C.c_int(-1), C.c_int(-1)
        self._check(self._lib.eth_kzg_amd_comm_info(self._ctx, C.byref(r), C.byref(w)))
        return r.value, w.value

    def all_gather(self, d_send, d_recv, bytes_per_rank, stream=None):
        """ncclAllGather of bytes_per_rank bytes from every rank's d_send into d_recv (device addresses), on `stream`
        (None: the context's stream, synchronised)."""
        self._check(self._lib.eth_kzg_amd_all_gather(self._ctx, C.c_void_p(d_send), C.c_void_p(d_recv), int(bytes_per_rank),
                                                     C.c_void_p(stream) if stream else None))

    def comm_destroy(self):
        self._lib.eth_kzg_amd_comm_destroy(self._ctx)

    @staticmethod
    def compute_cells_and_kzg_proofs_batch_multi(contexts, blobs_np, bufs):
        """eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi: one host-pointer batch cut into contiguous slices over
        `contexts` (one per GPU), buffers as in compute_cells_and_kzg_proofs_batch_np."""
        lib = load_library()
        n = blobs_np.shape[0]
        ba = np.uint64(blobs_np.ctypes.data) + np.arange(max(1, n), dtype=np.uint64) * np.uint64(BYTES_PER_BLOB)
        ctxs = (C.c_void_p * len(contexts))(*[c.handle for c in contexts])
        contexts[0]._check(lib.eth_kzg_amd_compute_cells_and_kzg_proofs_batch_multi(
            ctxs, len(contexts), n, _vp(ba), _vp(bufs["cpp"]), _vp(bufs["ppp"]), bufs["status"]))
        return list(bufs["status"])[:n]

    def verify_cell_kzg_proof_batch_device(self, n, d_commitments, d_cell_indices, d_cells, d_proofs, stream=None):
        """Device-resident verification: integer device addresses of n*48 commitment bytes (one per cell), n uint64 indices,
        n*2048 cell bytes, n*48 proof bytes."""
        ok = C.c_bool(False)
        self._check(self._lib.eth_kzg_amd_verify_cell_kzg_proof_batch_device(
            self._ctx, int(n), C.c_void_p(d_commitments), C.c_void_p(d_cell_indices), C.c_void_p(d_cells), C.c_void_p(d_proofs),
            C.byref(ok), C.c_void_p(stream) if stream else None))
        return bool(ok.value)

    def blob_to_kzg_commitment_device(self, n, d_blobs, d_out, want_status=True, stream=None):
        st = (C.c_int32 * max(1, n))() if want_status else None
        self._check(self._lib.eth_kzg_amd_blob_to_kzg_commitment_device(
            self._ctx, n, C.c_void_p(d_blobs), C.c_void_p(d_out), st, C.c_void_p(stream) if stream else None))
        return list(st)[:n] if want_status else None

    STAGES = ["blob_to_coeffs", "coeffs_to_cells", "fk20_scalars", "msm_fixed", "g1_ifft", "g1_fft", "compress", "g1_linmap"]

    def set_profiling(self, on):
        self._lib.eth_kzg_amd_set_profiling(self._ctx, int(bool(on)))

    def get_stage_times(self):
        """{stage: (milliseconds, kernel launches)} accumulated since the previous call."""
        n = len(self.STAGES)
        ms = (C.c_double * n)()
        ln = (C.c_uint64 * n)()
        self._lib.eth_kzg_amd_get_stage_times(self._ctx, ms, ln, n)
        return {s: (ms[i], int(ln[i])) for i, s in enumerate(self.STAGES)}

    def table_bytes(self):
        return int(self._lib.eth_kzg_amd_table_bytes(self._ctx))

    def window_bits(self):
        return int(self._lib.eth_kzg_amd_window_bits(self._ctx))

    def window_count(self):
        """Windows per 128-bit GLV half of the FK20 table in use (gathered additions per base = twice that)."""
        return int(self._lib.eth_kzg_amd_window_count(self._ctx))

    def linmap_info(self):
        """(constant multiplications, additions, doublings per blob, launches per call) of the compiled G1 linear map."""
        out = (C.c_int32 * 4)()
        self._lib.eth_kzg_amd_linmap_info(self._ctx, out)
        return tuple(out)
