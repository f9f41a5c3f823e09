// SHA-256 for the Fiat-Shamir transcript of the cell-proof batch verifier
// (reference: sha2 crate, crates/cryptography/kzg_multi_open/src/fk20/verifier.rs:269-328).
// Strictly sequential, so it stays on the host; uses the x86 SHA extensions when the CPU has them.
#pragma once
#include <cstddef>
#include <cstdint>
namespace kzg {
struct Sha256 {
    uint32_t h[8];
    uint8_t buf[64];
    size_t buf_len;
    uint64_t total;
    Sha256();
    void update(const uint8_t* data, size_t len);
    void finish(uint8_t out[32]);
};
bool sha256_uses_shani();
}  // namespace kzg
