// fv_capi.hip -- extern "C" entry points of libfftvis_hip.so (see include/fftvis_hip.h).
// Single translation unit: hipcc --offload-arch=gfx950 -O3 -shared -fPIC ... -lrocfft

#include "../../include/fftvis_hip.h"
#include "fv_sim.h"

#include <mutex>

namespace fv {

static thread_local std::string g_last_error;

void ensure_rocfft() {
    static std::once_flag once;
    std::call_once(once, [] {
        if (rocfft_setup() != rocfft_status_success)
            throw Error(FV_ERR_ROCFFT, "rocfft_setup failed");
    });
}

template <typename F>
static int guarded(F &&f) {
    try {
        f();
        return FV_OK;
    } catch (const Error &e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return FV_ERR_INTERNAL;
    } catch (...) {
        g_last_error = "unknown error";
        return FV_ERR_INTERNAL;
    }
}

struct StreamGuard {
    hipStream_t s = nullptr;
    StreamGuard() { FV_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~StreamGuard() {
        if (s) (void)hipStreamDestroy(s);
    }
};

static void minmax(const double *v, int64_t n, double &c, double &h) {
    double lo = 1e300, hi = -1e300;
    for (int64_t i = 0; i < n; ++i) {
        lo = std::min(lo, v[i]);
        hi = std::max(hi, v[i]);
    }
    if (n == 0) lo = hi = 0;
    c = 0.5 * (lo + hi);
    h = 0.5 * (hi - lo);
    // guard the box against the T-rounding of the uploaded coordinates
    h = h * (1.0 + 1e-6) + 1e-300;
}

template <typename T>
static void nufft3_host(int device, int dim, int64_t M, const void *const xin[3], const void *cin,
                        int ntrans, int64_t N, const void *const sin_[3], double eps,
                        double upsampfac, void *out, bool direct) {
    FV_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
    FV_REQUIRE(M >= 0 && N >= 0 && ntrans >= 1, "negative sizes");
    for (int d = 0; d < dim; ++d)
        FV_REQUIRE((xin[d] || M == 0) && (sin_[d] || N == 0), "missing coordinate array");
    FV_REQUIRE(cin || M == 0, "missing strengths");
    FV_REQUIRE(out || N == 0, "missing output");
    FV_HIP(hipSetDevice(device));
    if (N == 0) return;
    if (M == 0) {
        std::memset(out, 0, sizeof(cplx<T>) * (size_t)ntrans * N);
        return;
    }
    StreamGuard sg;
    hipStream_t st = sg.s;
    DevBuf dx[3], ds[3], dc, dout, dscale;
    double xc[3] = {0, 0, 0}, X[3] = {0, 0, 0}, sc[3] = {0, 0, 0}, S[3] = {0, 0, 0};
    std::vector<double> tmp;
    for (int d = 0; d < dim; ++d) {
        tmp.resize(std::max(M, N));
        const T *xv = (const T *)xin[d];
        for (int64_t i = 0; i < M; ++i) tmp[i] = xv[i];
        minmax(tmp.data(), M, xc[d], X[d]);
        const T *sv = (const T *)sin_[d];
        for (int64_t i = 0; i < N; ++i) tmp[i] = sv[i];
        minmax(tmp.data(), N, sc[d], S[d]);
        dx[d].reserve(sizeof(T) * M);
        ds[d].reserve(sizeof(T) * N);
        FV_HIP(hipMemcpyAsync(dx[d].p, xin[d], sizeof(T) * M, hipMemcpyHostToDevice, st));
        FV_HIP(hipMemcpyAsync(ds[d].p, sin_[d], sizeof(T) * N, hipMemcpyHostToDevice, st));
    }
    dc.reserve(sizeof(cplx<T>) * (size_t)ntrans * M);
    dout.reserve(sizeof(cplx<T>) * (size_t)ntrans * N);
    FV_HIP(hipMemcpyAsync(dc.p, cin, sizeof(cplx<T>) * (size_t)ntrans * M, hipMemcpyHostToDevice, st));

    if (direct) {
        hipLaunchKernelGGL(k_nudft_direct<T>, dim3((unsigned)N, (unsigned)ntrans), dim3(256), 0, st,
                           dim, M, dx[0].as<T>(), dx[1].as<T>(), dx[2].as<T>(), dc.as<cplx<T>>(),
                           ntrans, N, ds[0].as<T>(), ds[1].as<T>(), ds[2].as<T>(),
                           dout.as<cplx<T>>());
    } else {
        const double one = 1.0;
        dscale.reserve(sizeof(double));
        FV_HIP(hipMemcpyAsync(dscale.p, &one, sizeof(double), hipMemcpyHostToDevice, st));
        Nufft3<T> plan(dim, eps, upsampfac, st);
        plan.set_geometry(xc, X, sc, S, 1.0);
        plan.set_sources(M, dx[0].as<T>(), dx[1].as<T>(), dx[2].as<T>());
        plan.load_strengths(dc.as<cplx<T>>(), ntrans, ntrans, dscale.as<double>());
        plan.spread(ntrans);
        plan.fft(ntrans);
        int64_t pol_off[16];
        for (int r = 0; r < 16; ++r) pol_off[r] = (int64_t)r * N;
        plan.interp(N, ds[0].as<T>(), ds[1].as<T>(), ds[2].as<T>(), nullptr, nullptr,
                    dscale.as<double>(), 1, ntrans, dout.as<cplx<T>>(), 0, 1, pol_off, false);
        FV_HIP(hipMemcpyAsync(out, dout.p, sizeof(cplx<T>) * (size_t)ntrans * N,
                              hipMemcpyDeviceToHost, st));
        FV_HIP(hipStreamSynchronize(st));
        FV_HIP(hipGetLastError());
        return;
    }
    FV_HIP(hipMemcpyAsync(out, dout.p, sizeof(cplx<T>) * (size_t)ntrans * N, hipMemcpyDeviceToHost, st));
    FV_HIP(hipStreamSynchronize(st));
    FV_HIP(hipGetLastError());
}

}  // namespace fv

using namespace fv;

struct fv_sim {
    std::unique_ptr<SimBase> impl;
    int precision;
};

extern "C" {

int fv_version(void) { return 100; /* 0.1.0 */ }

int fv_device_count(int *count) {
    return guarded([&] {
        FV_REQUIRE(count, "null count");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            n = 0;
        }
        *count = n;
    });
}

const char *fv_last_error(void) { return g_last_error.c_str(); }

int fv_nufft3(int device, int precision, int dim, int64_t M, const void *x, const void *y,
              const void *z, const void *c, int ntrans, int64_t N, const void *s, const void *t,
              const void *u, double eps, double upsampfac, void *out) {
    return guarded([&] {
        FV_REQUIRE(precision == 1 || precision == 2, "precision must be 1 or 2");
        const void *xs[3] = {x, y, z}, *ss[3] = {s, t, u};
        if (precision == 2)
            nufft3_host<double>(device, dim, M, xs, c, ntrans, N, ss, eps, upsampfac, out, false);
        else
            nufft3_host<float>(device, dim, M, xs, c, ntrans, N, ss, eps, upsampfac, out, false);
    });
}

int fv_nudft3_direct(int device, int precision, int dim, int64_t M, const void *x, const void *y,
                     const void *z, const void *c, int ntrans, int64_t N, const void *s,
                     const void *t, const void *u, void *out) {
    return guarded([&] {
        FV_REQUIRE(precision == 1 || precision == 2, "precision must be 1 or 2");
        const void *xs[3] = {x, y, z}, *ss[3] = {s, t, u};
        if (precision == 2)
            nufft3_host<double>(device, dim, M, xs, c, ntrans, N, ss, 0, 2.0, out, true);
        else
            nufft3_host<float>(device, dim, M, xs, c, ntrans, N, ss, 0, 2.0, out, true);
    });
}

int fv_sim_create(fv_sim **h, int device, int precision, double eps, double upsampfac,
                  int polarized) {
    return guarded([&] {
        FV_REQUIRE(h, "null handle pointer");
        *h = nullptr;
        FV_REQUIRE(precision == 1 || precision == 2, "precision must be 1 or 2");
        FV_REQUIRE(upsampfac == 2.0 || upsampfac == 1.25, "upsample factor must be 2 or 1.25");
        FV_REQUIRE(eps > 0 && eps < 1, "eps must be in (0, 1)");
        std::unique_ptr<fv_sim> s(new fv_sim());
        s->precision = precision;
        if (precision == 2)
            s->impl.reset(new Sim<double>(device, eps, upsampfac, polarized));
        else
            s->impl.reset(new Sim<float>(device, eps, upsampfac, polarized));
        *h = s.release();
    });
}

int fv_sim_destroy(fv_sim *h) {
    return guarded([&] { delete h; });
}

#define FV_SIM_CALL(body)                      \
    return guarded([&] {                       \
        FV_REQUIRE(h && h->impl, "null handle"); \
        body;                                  \
    })

int fv_sim_set_sources(fv_sim *h, int64_t nsrc, int nfreq, const void *eq, const void *flux,
                       int polarized_sky, int on_device) {
    FV_SIM_CALL(h->impl->set_sources(nsrc, nfreq, eq, flux, polarized_sky, on_device));
}
int fv_sim_set_times(fv_sim *h, int ntimes, const double *rot) {
    FV_SIM_CALL(FV_REQUIRE(ntimes >= 0 && (rot || !ntimes), "bad times"); h->impl->set_times(ntimes, rot));
}
int fv_sim_set_freqs(fv_sim *h, int nfreq, const double *freqs) {
    FV_SIM_CALL(FV_REQUIRE(nfreq >= 1 && freqs, "bad freqs"); h->impl->set_freqs(nfreq, freqs));
}
int fv_sim_set_array(fv_sim *h, const double *R, int64_t nbls, const double *bls, int is_coplanar) {
    FV_SIM_CALL(FV_REQUIRE(R && nbls >= 1 && bls, "bad array"); h->impl->set_array(R, nbls, bls, is_coplanar));
}
int fv_sim_set_nbeams(fv_sim *h, int nbeams) {
    FV_SIM_CALL(FV_REQUIRE(nbeams >= 1, "need at least one beam"); h->impl->set_nbeams(nbeams));
}
int fv_sim_set_beam_airy(fv_sim *h, int beam, double diameter) {
    FV_SIM_CALL(FV_REQUIRE(diameter > 0, "diameter must be positive"); h->impl->set_beam_airy(beam, diameter));
}
int fv_sim_set_beam_table(fv_sim *h, int beam, int nfreq_tab, int nza, int naz, double za_max,
                          const void *table) {
    FV_SIM_CALL(FV_REQUIRE(table, "null table"); h->impl->set_beam_table(beam, nfreq_tab, nza, naz, za_max, table));
}
int fv_sim_set_beam_pairs(fv_sim *h, int npairs, const int *bi, const int *bj, const int64_t *off,
                          const int *idx, const signed char *flipped) {
    FV_SIM_CALL(FV_REQUIRE(npairs >= 1 && bi && bj && off, "bad pairs"); h->impl->set_beam_pairs(npairs, bi, bj, off, idx, flipped));
}
int fv_sim_run(fv_sim *h, int t0, int t1, int f0, int f1, void *out, int out_on_device) {
    FV_SIM_CALL(FV_REQUIRE(out, "null output"); h->impl->run(t0, t1, f0, f1, out, out_on_device));
}
int fv_sim_sync(fv_sim *h) { FV_SIM_CALL(h->impl->sync()); }
int fv_sim_stats(fv_sim *h, double *vals, int n) { FV_SIM_CALL(h->impl->stats(vals, n)); }
int fv_sim_reset_stats(fv_sim *h) { FV_SIM_CALL(h->impl->reset_stats()); }
int fv_sim_enable_timing(fv_sim *h, int on) { FV_SIM_CALL(h->impl->enable_timing(on)); }
int fv_sim_timing(fv_sim *h, double *ms, int n) { FV_SIM_CALL(h->impl->timing(ms, n)); }

}  // extern "C"
