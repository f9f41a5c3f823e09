// The object behind the C ABI's opaque `DASContext*` (include/c_eth_kzg.h), shared by c_api.cpp and multi_gpu.cpp.
// A context owns ONE ENGINE PER DEVICE of its device list (one device unless ETH_KZG_AMD_DEVICES or
// eth_kzg_amd_das_context_new_on_devices says otherwise): the reference's hosts create one context and share it between their
// threads (bindings/c/src/lib.rs:79-92, bindings/node/src/lib.rs:35,75), so the device list lives behind that one pointer.
#pragma once
#include "engine.hpp"
#include "host_sync.hpp"

#include <memory>
#include <vector>

struct DASContext {
    kzg::Engine* engine = nullptr;       // engines[0]: communicator, table queries, profiling, and every call of a one-device context
    std::vector<kzg::Engine*> engines;   // one per device of the list, in list order
    std::unique_ptr<kzg::DevicePicker> picker;  // single calls go to the least-loaded device (host_sync.hpp)
};
