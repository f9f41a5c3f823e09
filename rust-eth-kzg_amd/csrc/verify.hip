// placeholder translation unit: verify / recover paths (filled in next)
#include "engine.hpp"
namespace kzg {
int Engine::verify_cell_kzg_proof_batch_host(uint64_t, const uint8_t* const*, uint64_t, const uint64_t*, uint64_t,
                                             const uint8_t* const*, uint64_t, const uint8_t* const*, int*) {
    err_ = "verify path not built yet";
    return ERR_DEVICE;
}
int Engine::recover_cells_and_kzg_proofs_host(uint64_t, const uint8_t* const*, uint64_t, const uint64_t*, uint8_t* const*,
                                              uint8_t* const*) {
    err_ = "recover path not built yet";
    return ERR_DEVICE;
}
}  // namespace kzg
