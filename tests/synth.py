"""Deterministic synthetic inputs."""
import hashlib

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
SEED = b"\x4b\x5a\x47"


def seeded_blob(index: int, seed: bytes = SEED) -> bytes:
    """element k = SHA-256(seed || u32 blob || u32 k) mod r, 4096 elements, big-endian."""
    out = bytearray()
    for k in range(4096):
        h = hashlib.sha256(seed + index.to_bytes(4, "big") + k.to_bytes(4, "big")).digest()
        out += (int.from_bytes(h, "big") % R).to_bytes(32, "big")
    return bytes(out)


def dummy_blob() -> bytes:
    """blob_i = BE32(-i mod r): the reference's own bench input (crates/eip7594/benches/benchmark-mt.rs:10-17)."""
    return b"".join(((-i) % R).to_bytes(32, "big") for i in range(4096))


def seeded_scalars(n: int, tag: bytes, modulus: int = R, width: int = 32):
    return [(int.from_bytes(hashlib.sha512(tag + i.to_bytes(4, "big")).digest(), "big") % modulus).to_bytes(width, "big")
            for i in range(n)]
