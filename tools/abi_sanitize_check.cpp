// Host-side sanitizer check of libfftvis_hip's C ABI (ASan / UBSan on the CPU build).
// Built and run by `make -C oracle sanitize` against a copy of the library whose HOST code is compiled with
// -fsanitize=address,undefined (device code is not instrumented: GPU sanitizers are unavailable on this pool).
// Exercises every entry point's argument checking and error reporting -- the paths that run before any HIP
// call -- so it needs no GPU; a box with one also passes (calls that reach the runtime then succeed or
// report FV_ERR_HIP, both accepted where noted).
#include "../include/fftvis_hip.h"

#include <cstdio>
#include <cstring>
#include <vector>

static int fails = 0;
#define EXPECT(cond)                                                     \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond);  \
            ++fails;                                                     \
        }                                                                \
    } while (0)

int main() {
    EXPECT(fv_version() >= 100);
    int ndev = -1;
    EXPECT(fv_device_count(&ndev) == FV_OK && ndev >= 0);
    EXPECT(fv_device_count(nullptr) == FV_ERR_ARG);
    EXPECT(std::strlen(fv_last_error()) > 0);
    int64_t b = -1;
    EXPECT(fv_device_bytes(&b) == 0 && b >= 0);
    EXPECT(fv_device_bytes(nullptr) != 0);
    EXPECT(fv_device_mem_info(0, nullptr, nullptr) == FV_ERR_ARG);

    fv_sim *h = nullptr;
    EXPECT(fv_sim_create(nullptr, 0, 2, 1e-6, 2.0, 0) == FV_ERR_ARG);
    EXPECT(fv_sim_create(&h, 0, 3, 1e-6, 2.0, 0) == FV_ERR_ARG && h == nullptr);
    EXPECT(fv_sim_create(&h, 0, 2, 0.0, 2.0, 0) == FV_ERR_ARG && h == nullptr);
    EXPECT(fv_sim_create(&h, 0, 2, 1e-6, 1.7, 0) == FV_ERR_ARG && h == nullptr);
    EXPECT(std::strstr(fv_last_error(), "upsample") != nullptr);
    // every handle call refuses a null handle before touching it
    double d9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, f[2] = {1e8, 2e8}, v[16] = {0};
    int i2[2] = {0, 0};
    int64_t off[2] = {0, 1};
    signed char fl[1] = {0};
    EXPECT(fv_sim_set_sources(nullptr, 1, 1, d9, d9, 0, 0) == FV_ERR_ARG);
    EXPECT(fv_sim_set_times(nullptr, 1, d9) == FV_ERR_ARG);
    EXPECT(fv_sim_set_topo(nullptr, 1, 1, d9, 0) == FV_ERR_ARG);
    EXPECT(fv_sim_set_freqs(nullptr, 2, f) == FV_ERR_ARG);
    EXPECT(fv_sim_set_array(nullptr, d9, 1, d9, 1) == FV_ERR_ARG);
    EXPECT(fv_sim_set_array_type1(nullptr, d9, 1, i2, 1) == FV_ERR_ARG);
    EXPECT(fv_sim_set_nbeams(nullptr, 1) == FV_ERR_ARG);
    EXPECT(fv_sim_set_beam_airy(nullptr, 0, 14.0) == FV_ERR_ARG);
    EXPECT(fv_sim_set_beam_table(nullptr, 0, 1, 2, 2, 3.14, d9, 1) == FV_ERR_ARG);
    EXPECT(fv_sim_set_beam_pairs(nullptr, 1, i2, i2, off, i2, fl) == FV_ERR_ARG);
    EXPECT(fv_sim_set_basis(nullptr, 1, 1, 1, d9, i2, i2) == FV_ERR_ARG);
    EXPECT(fv_sim_set_chunking(nullptr, 1, 1.0) == FV_ERR_ARG);
    EXPECT(fv_sim_set_reference_compat(nullptr, 0) == FV_ERR_ARG);
    EXPECT(fv_sim_set_astrom(nullptr, 1, d9) == FV_ERR_ARG);
    EXPECT(fv_astrom_topo(0, 9, d9, 1, d9, v) == FV_ERR_ARG);
    EXPECT(fv_sim_set_beam_airy_scaled(nullptr, 0, 14.0, d9, 1.0) == FV_ERR_ARG);
    EXPECT(fv_sim_run(nullptr, 0, 1, 0, 1, v, 0) == FV_ERR_ARG);
    EXPECT(fv_sim_sync(nullptr) == FV_ERR_ARG);
    EXPECT(fv_sim_stats(nullptr, v, 12) == FV_ERR_ARG);
    EXPECT(fv_sim_reset_stats(nullptr) == FV_ERR_ARG);
    EXPECT(fv_sim_enable_timing(nullptr, 1) == FV_ERR_ARG);
    EXPECT(fv_sim_timing(nullptr, v, 6) == FV_ERR_ARG);
    EXPECT(fv_sim_destroy(nullptr) == FV_OK);  // delete nullptr
    // stand-alone ops: argument errors come before the device is touched
    std::vector<double> x(8, 0.5), c(16, 1.0), s(4, 2.0), out(64, 0.0);
    EXPECT(fv_nufft3(0, 3, 2, 8, x.data(), x.data(), nullptr, c.data(), 1, 4, s.data(), s.data(), nullptr, 1e-6, 2.0,
                     out.data()) == FV_ERR_ARG);
    EXPECT(fv_nufft3(0, 2, 4, 8, x.data(), x.data(), nullptr, c.data(), 1, 4, s.data(), s.data(), nullptr, 1e-6, 2.0,
                     out.data()) == FV_ERR_ARG);
    EXPECT(fv_nufft3(0, 2, 2, -1, x.data(), x.data(), nullptr, c.data(), 1, 4, s.data(), s.data(), nullptr, 1e-6, 2.0,
                     out.data()) == FV_ERR_ARG);
    EXPECT(fv_nufft3(0, 2, 2, 8, x.data(), nullptr, nullptr, c.data(), 1, 4, s.data(), s.data(), nullptr, 1e-6, 2.0,
                     out.data()) == FV_ERR_ARG);
    EXPECT(fv_nudft3_direct(0, 0, 2, 8, x.data(), x.data(), nullptr, c.data(), 1, 4, s.data(), s.data(), nullptr,
                            out.data()) == FV_ERR_ARG);
    EXPECT(fv_beam_eval(0, 2, 1, 2, 14.0, 0, 0, 0, 0.0, nullptr, 1, 0, 1.5e8, 4, x.data(), x.data(), out.data()) ==
           FV_ERR_ARG);
    EXPECT(fv_beam_eval(0, 2, 1, 0, 14.0, 0, 0, 0, 0.0, nullptr, 6, 0, 1.5e8, 4, x.data(), x.data(), out.data()) ==
           FV_ERR_ARG);
    EXPECT(fv_apparent_coherency(0, 2, 7, 4, c.data(), c.data(), x.data(), out.data()) == FV_ERR_ARG);
    EXPECT(fv_apparent_coherency(0, 5, 0, 4, c.data(), c.data(), x.data(), out.data()) == FV_ERR_ARG);
    EXPECT(fv_inplace_rot(0, 2, nullptr, x.data(), 2) == FV_ERR_ARG);
    EXPECT(fv_inplace_rot(0, 9, d9, x.data(), 2) == FV_ERR_ARG);
    // catalog exchange: argument checks come before RCCL is opened
    {
        fv_comm *cm = reinterpret_cast<fv_comm *>(0x1);
        unsigned char id[FV_COMM_ID_BYTES] = {0};
        EXPECT(fv_comm_unique_id(nullptr) == FV_ERR_ARG);
        EXPECT(fv_comm_init(nullptr, 0, 0, 1, id) == FV_ERR_ARG);
        EXPECT(fv_comm_init(&cm, 0, 0, 1, nullptr) == FV_ERR_ARG);
        EXPECT(fv_comm_init(&cm, 0, 2, 2, id) == FV_ERR_ARG);
        EXPECT(fv_comm_destroy(nullptr) == FV_OK);
        EXPECT(fv_bcast_catalog(nullptr, 0, v, 8, v, 8) == FV_ERR_ARG);
        EXPECT(fv_scatter_flux_columns(nullptr, 0, 1, 1, 8, v, i2, v) == FV_ERR_ARG);
    }
    EXPECT(fv_release_workspaces() == FV_OK);
    // a call that gets past the argument checks reports the missing device as a HIP error, not a crash
    if (ndev == 0) {
        EXPECT(fv_sim_create(&h, 0, 2, 1e-6, 2.0, 1) == FV_ERR_HIP && h == nullptr);
        EXPECT(fv_nufft3(0, 2, 2, 8, x.data(), x.data(), nullptr, c.data(), 1, 4, s.data(), s.data(), nullptr, 1e-6, 2.0,
                         out.data()) == FV_ERR_HIP);
    }
    std::printf(fails ? "abi_sanitize_check: %d failure(s)\n" : "abi_sanitize_check: ok\n", fails);
    return fails ? 1 : 0;
}
