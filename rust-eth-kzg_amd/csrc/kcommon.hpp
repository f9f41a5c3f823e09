// Shared definitions for the HIP kernel translation units (see launch.hpp for the host-side launchers).
#pragma once
#include "curve.hpp"

namespace kzg {

constexpr int N_BLOB = 4096;       // FIELD_ELEMENTS_PER_BLOB   (crates/serialization/src/constants.rs:9-65)
constexpr int N_EXT = 8192;        // FIELD_ELEMENTS_PER_EXT_BLOB
constexpr int N_CELLS = 128;       // CELLS_PER_EXT_BLOB
constexpr int CELL_LEN = 64;       // FIELD_ELEMENTS_PER_CELL
constexpr int BYTES_PER_BLOB = 131072;
constexpr int BYTES_PER_CELL = 2048;
constexpr size_t LDS_NTT = (size_t)N_BLOB * 32;  // 128 KiB: one 4096-point transform resident in LDS

// ------------------------------------------------------------------------------------------------
// byte codecs (serialization/src/lib.rs:36-63, 132-156)
__device__ __forceinline__ Fr load_fr_be(const uint8_t* p) {  // canonical integer, not Montgomery
    Fr r;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[7 - i] = __builtin_bswap32(w[i]);
    return r;
}
__device__ __forceinline__ void store_fr_be(uint8_t* p, const Fr& canon) {
    uint32_t* w = reinterpret_cast<uint32_t*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = __builtin_bswap32(canon.v[7 - i]);
}

}  // namespace kzg
