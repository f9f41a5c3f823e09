"""ctypes binding of oracle/liboracle.so -- the CPU oracle (test infrastructure only)."""
import ctypes as C
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "liboracle.so")

BYTES_PER_BLOB, BYTES_PER_CELL, CELLS, BYTES_PER_G1 = 131072, 2048, 128, 48


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle error {code}")
        self.code = code


def _lib():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
    lib = C.CDLL(_SO)
    lib.oracle_ctx_new.restype = C.c_void_p
    lib.oracle_ctx_new.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int]
    lib.oracle_ctx_free.argtypes = [C.c_void_p]
    return lib


class Oracle:
    def __init__(self, use_precomp=False, threads=1):
        self.lib = _lib()
        srs = open(os.path.join(_ROOT, "rust-eth-kzg_amd", "data", "trusted_setup_4096.bin"), "rb").read()
        self.ctx = C.c_void_p(self.lib.oracle_ctx_new(srs, len(srs), int(use_precomp), threads))
        assert self.ctx.value, "oracle_ctx_new failed"

    def close(self):
        if self.ctx:
            self.lib.oracle_ctx_free(self.ctx)
            self.ctx = None

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(rc)

    def blob_to_kzg_commitment(self, blob):
        if len(blob) != BYTES_PER_BLOB:
            raise OracleError(3)
        out = C.create_string_buffer(48)
        self._chk(self.lib.oracle_blob_to_kzg_commitment(self.ctx, blob, out))
        return out.raw

    def compute_cells_and_kzg_proofs(self, blob):
        if len(blob) != BYTES_PER_BLOB:
            raise OracleError(3)
        cells = C.create_string_buffer(CELLS * BYTES_PER_CELL)
        proofs = C.create_string_buffer(CELLS * 48)
        self._chk(self.lib.oracle_compute_cells_and_kzg_proofs(self.ctx, blob, cells, proofs))
        return ([cells.raw[i * 2048:(i + 1) * 2048] for i in range(CELLS)],
                [proofs.raw[i * 48:(i + 1) * 48] for i in range(CELLS)])

    def compute_cells(self, blob):
        if len(blob) != BYTES_PER_BLOB:
            raise OracleError(3)
        cells = C.create_string_buffer(CELLS * BYTES_PER_CELL)
        self._chk(self.lib.oracle_compute_cells(self.ctx, blob, cells))
        return [cells.raw[i * 2048:(i + 1) * 2048] for i in range(CELLS)]

    def verify_cell_kzg_proof_batch(self, commitments, cell_indices, cells, proofs):
        if any(len(c) != 48 for c in commitments) or any(len(p) != 48 for p in proofs) or \
                any(len(c) != BYTES_PER_CELL for c in cells):
            raise OracleError(3)
        idx = (C.c_uint64 * max(1, len(cell_indices)))(*cell_indices)
        ok = C.c_int(0)
        self._chk(self.lib.oracle_verify_cell_kzg_proof_batch(
            self.ctx, C.c_size_t(len(commitments)), b"".join(commitments), C.c_size_t(len(cell_indices)), idx,
            C.c_size_t(len(cells)), b"".join(cells), C.c_size_t(len(proofs)), b"".join(proofs), C.byref(ok)))
        return bool(ok.value)

    def recover_cells_and_kzg_proofs(self, cell_indices, cells):
        if any(len(c) != BYTES_PER_CELL for c in cells):
            raise OracleError(3)
        idx = (C.c_uint64 * max(1, len(cell_indices)))(*cell_indices)
        oc = C.create_string_buffer(CELLS * BYTES_PER_CELL)
        op = C.create_string_buffer(CELLS * 48)
        self._chk(self.lib.oracle_recover_cells_and_kzg_proofs(
            self.ctx, C.c_size_t(len(cells)), b"".join(cells), C.c_size_t(len(cell_indices)), idx, oc, op))
        return ([oc.raw[i * 2048:(i + 1) * 2048] for i in range(CELLS)],
                [op.raw[i * 48:(i + 1) * 48] for i in range(CELLS)])


# ---- stage-level oracle entry points (canonical encodings; see oracle/kzg_oracle.h) ----
def fr_ntt(data: bytes, inverse=False, coset=0) -> bytes:
    lib = _lib()
    n = len(data) // 32
    buf = C.create_string_buffer(data, len(data))
    rc = lib.oracle_fr_ntt(buf, C.c_size_t(n), int(inverse), int(coset))
    if rc:
        raise OracleError(rc)
    return buf.raw


def g1_fft(points: bytes, inverse=False) -> bytes:
    lib = _lib()
    n = len(points) // 48
    buf = C.create_string_buffer(points, len(points))
    rc = lib.oracle_g1_fft(buf, C.c_size_t(n), int(inverse))
    if rc:
        raise OracleError(rc)
    return buf.raw


def g1_msm(points: bytes, scalars: bytes) -> bytes:
    lib = _lib()
    n = len(points) // 48
    out = C.create_string_buffer(48)
    rc = lib.oracle_g1_msm(points, scalars, C.c_size_t(n), out)
    if rc:
        raise OracleError(rc)
    return out.raw


def g1_mul(point: bytes, scalar: bytes) -> bytes:
    out = C.create_string_buffer(48)
    rc = _lib().oracle_g1_mul(point, scalar, out)
    if rc:
        raise OracleError(rc)
    return out.raw


def g1_validate(point: bytes, subgroup_check=True) -> int:
    return _lib().oracle_g1_validate(point, int(subgroup_check))


def fr_mul(a: bytes, b: bytes) -> bytes:
    out = C.create_string_buffer(32)
    rc = _lib().oracle_fr_mul(a, b, out)
    if rc:
        raise OracleError(rc)
    return out.raw


def fp_mul(a: bytes, b: bytes) -> bytes:
    out = C.create_string_buffer(48)
    rc = _lib().oracle_fp_mul(a, b, out)
    if rc:
        raise OracleError(rc)
    return out.raw


def sha256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    _lib().oracle_sha256(data, C.c_size_t(len(data)), out)
    return out.raw


# ---- EIP-4844 single-point operations on the Oracle object ----
def _o_compute_kzg_proof(self, blob, z):
    if len(blob) != BYTES_PER_BLOB or len(z) != 32:
        raise OracleError(3)
    p, y = C.create_string_buffer(48), C.create_string_buffer(32)
    self._chk(self.lib.oracle_compute_kzg_proof(self.ctx, blob, z, p, y))
    return p.raw, y.raw


def _o_compute_blob_kzg_proof(self, blob, commitment):
    if len(blob) != BYTES_PER_BLOB or len(commitment) != 48:
        raise OracleError(3)
    p = C.create_string_buffer(48)
    self._chk(self.lib.oracle_compute_blob_kzg_proof(self.ctx, blob, commitment, p))
    return p.raw


def _o_verify_kzg_proof(self, commitment, z, y, proof):
    if len(commitment) != 48 or len(proof) != 48 or len(z) != 32 or len(y) != 32:
        raise OracleError(3)
    ok = C.c_int(0)
    self._chk(self.lib.oracle_verify_kzg_proof(self.ctx, commitment, z, y, proof, C.byref(ok)))
    return bool(ok.value)


def _o_verify_blob_kzg_proof(self, blob, commitment, proof):
    if len(blob) != BYTES_PER_BLOB or len(commitment) != 48 or len(proof) != 48:
        raise OracleError(3)
    ok = C.c_int(0)
    self._chk(self.lib.oracle_verify_blob_kzg_proof(self.ctx, blob, commitment, proof, C.byref(ok)))
    return bool(ok.value)


def _o_verify_blob_kzg_proof_batch(self, blobs, commitments, proofs):
    if any(len(b) != BYTES_PER_BLOB for b in blobs) or any(len(c) != 48 for c in commitments) or any(len(p) != 48 for p in proofs):
        raise OracleError(3)
    ok = C.c_int(0)
    self._chk(self.lib.oracle_verify_blob_kzg_proof_batch(
        self.ctx, C.c_size_t(len(blobs)), b"".join(blobs), C.c_size_t(len(commitments)), b"".join(commitments),
        C.c_size_t(len(proofs)), b"".join(proofs), C.byref(ok)))
    return bool(ok.value)


Oracle.compute_kzg_proof = _o_compute_kzg_proof
Oracle.compute_blob_kzg_proof = _o_compute_blob_kzg_proof
Oracle.verify_kzg_proof = _o_verify_kzg_proof
Oracle.verify_blob_kzg_proof = _o_verify_blob_kzg_proof
Oracle.verify_blob_kzg_proof_batch = _o_verify_blob_kzg_proof_batch
