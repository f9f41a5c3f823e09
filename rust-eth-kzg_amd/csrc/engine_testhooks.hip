// Stage-level test hooks of the engine (include/c_eth_kzg_test_hooks.h): single stages against the oracle.
#include "engine_internal.hpp"

namespace kzg {

// ---------------------------------------------------------------------------------------------
// stage-level test hooks
int Engine::test_fr_ntt4096(const uint8_t* in_be, uint8_t* out_be, int inverse_dit) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        uint8_t *di, *dout;
        HIPCK(hipMalloc(&di, BYTES_PER_BLOB));
        HIPCK(hipMalloc(&dout, BYTES_PER_BLOB));
        HIPCK(hipMemcpy(di, in_be, BYTES_PER_BLOB, hipMemcpyHostToDevice));
        launch::test_ntt4096(di, dout, d_w29_, n_inv4096_, inverse_dit, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        HIPCK(hipMemcpy(out_be, dout, BYTES_PER_BLOB, hipMemcpyDeviceToHost));
        HIPCK(hipFree(di));
        HIPCK(hipFree(dout));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// in/out: [lane][128][48 B]; both directions natural in -> natural out (inverse is unscaled)
int Engine::test_g1_fft128(const uint8_t* in, uint8_t* out, int n_lanes, int inverse) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        int stride = ((n_lanes + 63) / 64) * 64;
        size_t bytes = (size_t)n_lanes * 128 * 48;
        uint8_t *di, *dout;
        void* X;
        HIPCK(hipMalloc(&di, bytes));
        HIPCK(hipMalloc(&dout, bytes));
        size_t nx = (size_t)128 * stride;
        const size_t PS = launch::SIZEOF_JACQ;
        HIPCK(hipMalloc(&X, nx * PS));
        HIPCK(hipMemcpy(di, in, bytes, hipMemcpyHostToDevice));
        launch::g1_set_inf(X, nx, stream_);
        launch::test_load_points(di, X, n_lanes, stride, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        std::vector<uint8_t> hx(nx * PS), hy(nx * PS);
        auto brp = [](int v) { int r = 0; for (int i = 0; i < 7; i++) r |= ((v >> i) & 1) << (6 - i); return r; };
        auto permute = [&]() {
            HIPCK(hipMemcpy(hx.data(), X, nx * PS, hipMemcpyDeviceToHost));
            for (int p = 0; p < 128; p++) memcpy(&hy[(size_t)brp(p) * stride * PS], &hx[(size_t)p * stride * PS], stride * PS);
            HIPCK(hipMemcpy(X, hy.data(), nx * PS, hipMemcpyHostToDevice));
        };
        if (inverse) permute();  // DIT wants bit-reversed input
        g1_fft128_full(X, stride, inverse, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        if (!inverse) permute();  // DIF leaves bit-reversed output
        launch::g1_compress(X, dout, 128, stride, n_lanes, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        HIPCK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
        HIPCK(hipFree(di)); HIPCK(hipFree(dout)); HIPCK(hipFree(X));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

// scalars: [n_msm][128 groups][64] BE -> out [n_msm][128][48]: the 128 fixed-base MSMs of stage D
int Engine::test_fixed_msm(const uint8_t* scalars_be, int n_msm, uint8_t* out) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        size_t ns = (size_t)n_msm * 128 * 64;
        int stride = ((n_msm + 63) / 64) * 64;
        uint8_t *di, *dout;
        void *sc, *X;
        HIPCK(hipMalloc(&di, ns * 32));
        HIPCK(hipMalloc(&sc, ns * sizeof(Fr)));
        HIPCK(hipMalloc(&X, (size_t)128 * stride * launch::SIZEOF_JACQ));
        HIPCK(hipMalloc(&dout, (size_t)n_msm * 128 * 48));
        HIPCK(hipMemcpy(di, scalars_be, ns * 32, hipMemcpyHostToDevice));
        launch::test_scalars_be(di, sc, ns, stream_);
        launch::g1_set_inf(X, (size_t)128 * stride, stream_);
        launch_msm(sc, TAB_FK, X, 128, n_msm, stride, 0, stream_);
        launch::g1_compress(X, dout, 128, stride, n_msm, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        HIPCK(hipMemcpy(out, dout, (size_t)n_msm * 128 * 48, hipMemcpyDeviceToHost));
        HIPCK(hipFree(di)); HIPCK(hipFree(sc)); HIPCK(hipFree(X)); HIPCK(hipFree(dout));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

int Engine::test_g1_decompress(const uint8_t* in, int n, int subgroup_check, int* h_status, uint8_t* out) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        uint8_t *di, *dout;
        void* pts;
        int* st;
        HIPCK(hipMalloc(&di, (size_t)n * 48)); HIPCK(hipMalloc(&dout, (size_t)n * 48));
        HIPCK(hipMalloc(&pts, (size_t)n * sizeof(G1Affine))); HIPCK(hipMalloc(&st, n * sizeof(int)));
        HIPCK(hipMemcpy(di, in, (size_t)n * 48, hipMemcpyHostToDevice));
        launch::g1_decompress(di, pts, st, n, subgroup_check, beta_, stream_);
        launch::test_recompress(pts, dout, n, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        HIPCK(hipMemcpy(h_status, st, n * sizeof(int), hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(out, dout, (size_t)n * 48, hipMemcpyDeviceToHost));
        HIPCK(hipFree(di)); HIPCK(hipFree(dout)); HIPCK(hipFree(pts)); HIPCK(hipFree(st));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

int Engine::test_field_mul(const uint8_t* a, const uint8_t* b, uint8_t* out, int n, int is_fp) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    try {
        HIPCK(hipSetDevice(dev_));
        size_t nb = (size_t)n * (is_fp ? 48 : 32);
        uint8_t *da, *db, *dout;
        HIPCK(hipMalloc(&da, nb)); HIPCK(hipMalloc(&db, nb)); HIPCK(hipMalloc(&dout, nb));
        HIPCK(hipMemcpy(da, a, nb, hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(db, b, nb, hipMemcpyHostToDevice));
        launch::test_field_mul(da, db, dout, n, is_fp, stream_);
        HIPCK(hipStreamSynchronize(stream_));
        HIPCK(hipMemcpy(out, dout, nb, hipMemcpyDeviceToHost));
        HIPCK(hipFree(da)); HIPCK(hipFree(db)); HIPCK(hipFree(dout));
    } catch (const std::exception& e) {
        set_error(e);
        return ERR_DEVICE;
    }
    return OK;
}

}  // namespace kzg
