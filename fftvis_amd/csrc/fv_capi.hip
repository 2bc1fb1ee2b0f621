// fv_capi.hip -- extern "C" entry points of libfftvis_hip.so (see include/fftvis_hip.h).
// Single translation unit: hipcc --offload-arch=gfx950 -O3 -shared -fPIC ...

#include "../../include/fftvis_hip.h"
#include "fv_sim.h"

#include <algorithm>
#include <atomic>
#include <mutex>

namespace fv {

static thread_local std::string g_last_error;

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

// Stream, device buffers and plan of the stand-alone transform, kept per host thread between calls
// with the same (device, dim, eps, upsampling factor): setting them up costs ~5 ms, more than a small
// transform.  Held through plain pointers that only fv_release_workspaces() deletes (static
// destructors must not call into a HIP runtime that may be gone); dropped after a call that leaves the
// process above FFTVIS_HIP_HANDLE_CACHE_BYTES (default 2 GiB) of device memory.
template <typename T>
struct NufftWorkspace {
    int device = -1, dim = 0;
    double eps = 0, sigma = 0;
    uint64_t id = 0;  // unique over the life of the process: a thread's slot is (pointer, id), so a workspace that
                      // another thread freed and a NEW one that happens to sit at the same address never compare equal
    hipStream_t st = nullptr;
    DevBuf dx[3], ds[3], dc, dout, dscale;
    std::unique_ptr<Nufft3<T>> plan;
    ~NufftWorkspace() {
        plan.reset();
        if (st) (void)hipStreamDestroy(st);
    }
};
template <typename T>
static NufftWorkspace<T> *&workspace_slot() {
    static thread_local NufftWorkspace<T> *ws = nullptr;
    return ws;
}
template <typename T>
static uint64_t &workspace_slot_id() {
    static thread_local uint64_t id = 0;
    return id;
}
static uint64_t next_workspace_id() {
    static std::atomic<uint64_t> n{0};
    return ++n;
}
// Every live workspace, whichever thread made it, so that fv_release_workspaces() can free those of
// threads that have exited (their thread_local pointer died with them, the device memory did not).
// A thread's slot holds an index into this registry rather than ownership.
template <typename T>
struct WorkspaceRegistry {
    std::mutex mu;
    std::vector<NufftWorkspace<T> *> all;
    static WorkspaceRegistry &get() {
        static WorkspaceRegistry *r = new WorkspaceRegistry();  // never destroyed: see the note above
        return *r;
    }
    void add(NufftWorkspace<T> *w) {
        std::lock_guard<std::mutex> g(mu);
        all.push_back(w);
    }
    bool remove(NufftWorkspace<T> *w) {  // true if it was still registered (i.e. nobody freed it yet)
        std::lock_guard<std::mutex> g(mu);
        auto it = std::find(all.begin(), all.end(), w);
        if (it == all.end()) return false;
        all.erase(it);
        return true;
    }
    bool contains(NufftWorkspace<T> *w, uint64_t id) {  // registered AND the same object the slot was made for
        std::lock_guard<std::mutex> g(mu);
        auto it = std::find(all.begin(), all.end(), w);
        return it != all.end() && (*it)->id == id;
    }
    bool remove(NufftWorkspace<T> *w, uint64_t id) {
        std::lock_guard<std::mutex> g(mu);
        auto it = std::find(all.begin(), all.end(), w);
        if (it == all.end() || (*it)->id != id) return false;
        all.erase(it);
        return true;
    }
    void drain() {
        std::vector<NufftWorkspace<T> *> take;
        {
            std::lock_guard<std::mutex> g(mu);
            take.swap(all);
        }
        for (NufftWorkspace<T> *w : take) {
            (void)hipSetDevice(w->device);
            delete w;
        }
    }
};
static size_t workspace_limit() {
    const char *e = std::getenv("FFTVIS_HIP_HANDLE_CACHE_BYTES");
    return e ? (size_t)std::atof(e) : ((size_t)2 << 30);
}
template <typename T>
static void drop_workspace() {
    NufftWorkspace<T> *&ws = workspace_slot<T>();
    if (ws) {
        if (WorkspaceRegistry<T>::get().remove(ws, workspace_slot_id<T>())) {  // else another thread's fv_release_workspaces() freed it
            (void)hipSetDevice(ws->device);
            delete ws;
        }
        ws = nullptr;
    }
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
    NufftWorkspace<T> *&slot = workspace_slot<T>();
    if (slot && !WorkspaceRegistry<T>::get().contains(slot, workspace_slot_id<T>())) slot = nullptr;  // freed by another thread's release
    if (slot && (slot->device != device || slot->dim != dim || slot->eps != eps || slot->sigma != upsampfac))
        drop_workspace<T>();
    if (!slot) {
        slot = new NufftWorkspace<T>();
        slot->device = device;
        slot->dim = dim;
        slot->eps = eps;
        slot->sigma = upsampfac;
        slot->id = workspace_slot_id<T>() = next_workspace_id();
        FV_HIP(hipStreamCreateWithFlags(&slot->st, hipStreamNonBlocking));
        WorkspaceRegistry<T>::get().add(slot);
    }
    struct DropOnError {  // an exception may leave the plan half configured
        bool armed = true;
        ~DropOnError() {
            if (armed) drop_workspace<T>();
        }
    } guard;
    NufftWorkspace<T> &W = *slot;
    hipStream_t st = W.st;
    DevBuf(&dx)[3] = W.dx, (&ds)[3] = W.ds, &dc = W.dc, &dout = W.dout, &dscale = W.dscale;
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
        if (!W.plan) W.plan.reset(new Nufft3<T>(dim, eps, upsampfac, st));
        Nufft3<T> &plan = *W.plan;
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
        // finufft rejects points it cannot place; here they were clamped to an edge cell and counted
        const int noob = plan.out_of_box_count();
        FV_REQUIRE(noob == 0, std::to_string(noob) + " source coordinates are NaN or outside the box of the "
                                                       "finite ones: the transform is invalid");
        guard.armed = fv::dev_bytes_held().load() > workspace_limit();
        return;
    }
    FV_HIP(hipMemcpyAsync(out, dout.p, sizeof(cplx<T>) * (size_t)ntrans * N, hipMemcpyDeviceToHost, st));
    FV_HIP(hipStreamSynchronize(st));
    FV_HIP(hipGetLastError());
    guard.armed = fv::dev_bytes_held().load() > workspace_limit();
}

template <typename T>
static void beam_eval_host(int device, int polarized, int kind, double diameter, int nft, int nza,
                           int naz, double za_max, const void *table, int order, int fidx, double freq,
                           int64_t n, const void *az, const void *za, void *out) {
    FV_REQUIRE(kind == 0 || kind == 1, "beam kind must be 0 (Airy) or 1 (table)");
    FV_REQUIRE(order >= 0 && order <= 5, "beam interpolation order must be 0 .. 5");
    FV_REQUIRE(n >= 0 && (n == 0 || (az && za && out)), "bad beam_eval arrays");
    FV_HIP(hipSetDevice(device));
    if (n == 0) return;
    StreamGuard sg;
    DevBuf daz, dza, dout, dtab, dtab_in;
    BeamDesc b{};
    b.kind = kind;
    b.diameter = diameter;
    for (int i = 0; i < 8; ++i) b.js[i] = i % 2 ? 0.0 : 1.0;
    b.ps = 1.0;
    if (kind == 0 && table) {  // Airy with factors: 8 doubles (Jones slots, re / im) + the power factor
        const double *f = static_cast<const double *>(table);
        for (int i = 0; i < 8; ++i) b.js[i] = f[i];
        b.ps = f[8];
    }
    if (kind == 1) {
        FV_REQUIRE(table && nza >= 2 && naz >= 1 && nft >= 1 && za_max > 0, "bad beam table");
        FV_REQUIRE(nft == 1 || (fidx >= 0 && fidx < nft), "freq_index outside the beam table");
        const size_t bytes = (polarized ? 64 : 8) * (size_t)nft * nza * naz;
        dtab.reserve(bytes);
        if (!polarized) {
            FV_HIP(hipMemcpyAsync(dtab.p, table, bytes, hipMemcpyHostToDevice, sg.s));
        } else {  // Jones tables are read in the interleaved device layout (eval_jones)
            dtab_in.reserve(bytes);
            FV_HIP(hipMemcpyAsync(dtab_in.p, table, bytes, hipMemcpyHostToDevice, sg.s));
            const int64_t nodes = (int64_t)nza * naz;
            hipLaunchKernelGGL(k_jones_interleave, dim3((unsigned)cdiv(nodes * nft, 256)), dim3(256), 0, sg.s,
                               dtab_in.as<cplx<double>>(), dtab.as<cplx<double>>(), nodes, (int64_t)nft);
        }
        bspline_prefilter(dtab.as<double>(), nft, nza, naz, polarized ? 8 : 1, order, sg.s);
        b.order = order;
        b.table = dtab.p;
        b.nfreq_tab = nft;
        b.nza = nza;
        b.naz = naz;
        b.za_max = za_max;
    } else {
        FV_REQUIRE(diameter > 0, "diameter must be positive");
    }
    const size_t nout = (polarized ? 4 : 1) * (size_t)n;
    daz.reserve(sizeof(T) * n);
    dza.reserve(sizeof(T) * n);
    dout.reserve(sizeof(cplx<T>) * nout);
    FV_HIP(hipMemcpyAsync(daz.p, az, sizeof(T) * n, hipMemcpyHostToDevice, sg.s));
    FV_HIP(hipMemcpyAsync(dza.p, za, sizeof(T) * n, hipMemcpyHostToDevice, sg.s));
    hipLaunchKernelGGL((order == 3 ? k_beam_eval<T, 3> : order == 1 ? k_beam_eval<T, 1> : k_beam_eval<T, 0>), dim3(cdiv(n, 256)), dim3(256), 0, sg.s, b, polarized, fidx,
                       freq, n, daz.as<T>(), dza.as<T>(), dout.as<cplx<T>>());
    FV_HIP(hipMemcpyAsync(out, dout.p, sizeof(cplx<T>) * nout, hipMemcpyDeviceToHost, sg.s));
    FV_HIP(hipStreamSynchronize(sg.s));
    FV_HIP(hipGetLastError());
}

template <typename T>
static void coherency_host(int device, int variant, int64_t n, const void *bi, const void *bj,
                           const void *flux, void *out) {
    FV_REQUIRE(variant >= 0 && variant <= 4, "coherency variant must be 0..4");
    FV_REQUIRE(n >= 0 && (n == 0 || (bi && bj && flux && out)), "bad coherency arrays");
    FV_HIP(hipSetDevice(device));
    if (n == 0) return;
    StreamGuard sg;
    const size_t nb = (variant == 4 ? 1 : 4) * (size_t)n;
    const bool cflux = variant == 1 || variant == 3;
    const size_t fbytes = cflux ? sizeof(cplx<T>) * 4 * n : sizeof(T) * n;
    DevBuf dbi, dbj, dfl, dout;
    dbi.reserve(sizeof(cplx<T>) * nb);
    dbj.reserve(sizeof(cplx<T>) * nb);
    dfl.reserve(fbytes);
    dout.reserve(sizeof(cplx<T>) * nb);
    FV_HIP(hipMemcpyAsync(dbi.p, bi, sizeof(cplx<T>) * nb, hipMemcpyHostToDevice, sg.s));
    FV_HIP(hipMemcpyAsync(dbj.p, bj, sizeof(cplx<T>) * nb, hipMemcpyHostToDevice, sg.s));
    FV_HIP(hipMemcpyAsync(dfl.p, flux, fbytes, hipMemcpyHostToDevice, sg.s));
    hipLaunchKernelGGL(k_apparent_coherency<T>, dim3(cdiv(n, 256)), dim3(256), 0, sg.s, variant, n,
                       dbi.as<cplx<T>>(), dbj.as<cplx<T>>(), dfl.p, dout.as<cplx<T>>());
    FV_HIP(hipMemcpyAsync(out, dout.p, sizeof(cplx<T>) * nb, hipMemcpyDeviceToHost, sg.s));
    FV_HIP(hipStreamSynchronize(sg.s));
    FV_HIP(hipGetLastError());
}

template <typename T>
static void inplace_rot_host(int device, const double *rot, void *b, int64_t n) {
    FV_REQUIRE(rot && n >= 0 && (b || n == 0), "bad inplace_rot arrays");
    FV_HIP(hipSetDevice(device));
    if (n == 0) return;
    StreamGuard sg;
    DevBuf db;
    db.reserve(sizeof(T) * 3 * n);
    Rot9 r;
    std::memcpy(r.m, rot, sizeof(r.m));
    FV_HIP(hipMemcpyAsync(db.p, b, sizeof(T) * 3 * n, hipMemcpyHostToDevice, sg.s));
    hipLaunchKernelGGL(k_inplace_rot<T>, dim3(cdiv(n, 256)), dim3(256), 0, sg.s, r, db.as<T>(), n);
    FV_HIP(hipMemcpyAsync(b, db.p, sizeof(T) * 3 * n, hipMemcpyDeviceToHost, sg.s));
    FV_HIP(hipStreamSynchronize(sg.s));
    FV_HIP(hipGetLastError());
}

template <typename T>
static void astrom_topo_host(int device, const double *astrom, int64_t n, const void *eq, void *out) {
    FV_REQUIRE(astrom && n >= 0 && ((eq && out) || n == 0), "bad astrom_topo arrays");
    FV_HIP(hipSetDevice(device));
    if (n == 0) return;
    Astrom a;
    std::memcpy(&a, astrom, sizeof(a));
    FV_REQUIRE(a.em > 0 && a.bm1 > 0, "astrometry context: em and bm1 must be positive");
    StreamGuard sg;
    DevBuf din, dout;
    din.reserve(sizeof(T) * 3 * n);
    dout.reserve(sizeof(T) * 3 * n);
    FV_HIP(hipMemcpyAsync(din.p, eq, sizeof(T) * 3 * n, hipMemcpyHostToDevice, sg.s));
    hipLaunchKernelGGL(k_astrom_topo<T>, dim3(cdiv(n, 256)), dim3(256), 0, sg.s, n, n, (int64_t)0, din.as<T>(), a, dout.as<T>());
    FV_HIP(hipMemcpyAsync(out, dout.p, sizeof(T) * 3 * n, hipMemcpyDeviceToHost, sg.s));
    FV_HIP(hipStreamSynchronize(sg.s));
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
int fv_release_workspaces(void) {
    return guarded([&] {  // every thread's: call while no fv_nufft3 / fv_nudft3_direct is in flight
        workspace_slot<double>() = nullptr;
        workspace_slot<float>() = nullptr;
        WorkspaceRegistry<double>::get().drain();
        WorkspaceRegistry<float>::get().drain();
    });
}
int fv_device_bytes(int64_t *bytes) {
    if (bytes) *bytes = (int64_t)fv::dev_bytes_held().load();
    return bytes ? 0 : 1;
}

int fv_device_bytes_on(int device, int64_t *bytes) {
    if (bytes) *bytes = (int64_t)fv::dev_bytes_on(device).load();
    return bytes ? 0 : 1;
}

int fv_device_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes) {
    return guarded([&] {
        FV_REQUIRE(free_bytes && total_bytes, "null output");
        FV_HIP(hipSetDevice(device));
        size_t f = 0, t = 0;
        FV_HIP(hipMemGetInfo(&f, &t));
        *free_bytes = (int64_t)f;
        *total_bytes = (int64_t)t;
    });
}

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

int fv_beam_eval(int device, int precision, int polarized, int kind, double diameter,
                 int nfreq_tab, int nza, int naz, double za_max, const void *table, int order,
                 int freq_index, double freq, int64_t n, const void *az, const void *za, void *out) {
    return guarded([&] {
        FV_REQUIRE(precision == 1 || precision == 2, "precision must be 1 or 2");
        if (precision == 2)
            beam_eval_host<double>(device, polarized, kind, diameter, nfreq_tab, nza, naz, za_max,
                                   table, order, freq_index, freq, n, az, za, out);
        else
            beam_eval_host<float>(device, polarized, kind, diameter, nfreq_tab, nza, naz, za_max,
                                  table, order, freq_index, freq, n, az, za, out);
    });
}

int fv_apparent_coherency(int device, int precision, int variant, int64_t n, const void *beam_i,
                          const void *beam_j, const void *flux, void *out) {
    return guarded([&] {
        FV_REQUIRE(precision == 1 || precision == 2, "precision must be 1 or 2");
        if (precision == 2)
            coherency_host<double>(device, variant, n, beam_i, beam_j, flux, out);
        else
            coherency_host<float>(device, variant, n, beam_i, beam_j, flux, out);
    });
}

int fv_inplace_rot(int device, int precision, const double *rot, void *b, int64_t n) {
    return guarded([&] {
        FV_REQUIRE(precision == 1 || precision == 2, "precision must be 1 or 2");
        if (precision == 2)
            inplace_rot_host<double>(device, rot, b, n);
        else
            inplace_rot_host<float>(device, rot, b, n);
    });
}

int fv_astrom_topo(int device, int precision, const double *astrom, int64_t n, const void *eq, void *topo) {
    return guarded([&] {
        FV_REQUIRE(precision == 1 || precision == 2, "precision must be 1 or 2");
        if (precision == 2)
            astrom_topo_host<double>(device, astrom, n, eq, topo);
        else
            astrom_topo_host<float>(device, astrom, n, eq, topo);
    });
}

int fv_sim_create(fv_sim **h, int device, int precision, double eps, double upsampfac,
                  int polarized) {
    return guarded([&] {
        FV_REQUIRE(h, "null handle pointer");
        *h = nullptr;
        FV_REQUIRE(precision == 1 || precision == 2, "precision must be 1 or 2");
        FV_REQUIRE(upsampfac == 2.0 || upsampfac == 1.25 || upsampfac == 0.0, "upsample factor must be 2, 1.25 or 0 (auto)");
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
int fv_sim_set_astrom(fv_sim *h, int ntimes, const double *astrom) {
    FV_SIM_CALL(FV_REQUIRE(ntimes >= 1 && astrom, "bad astrometry contexts"); h->impl->set_astrom(ntimes, astrom));
}
int fv_sim_set_topo(fv_sim *h, int ntimes, int64_t nsrc, const void *topo, int on_device) {
    FV_SIM_CALL(FV_REQUIRE(ntimes >= 1 && topo, "bad topo"); h->impl->set_topo(ntimes, nsrc, topo, on_device));
}
int fv_sim_set_freqs(fv_sim *h, int nfreq, const double *freqs) {
    FV_SIM_CALL(FV_REQUIRE(nfreq >= 1 && freqs, "bad freqs"); h->impl->set_freqs(nfreq, freqs));
}
int fv_sim_set_array(fv_sim *h, const double *R, int64_t nbls, const double *bls, int is_coplanar) {
    FV_SIM_CALL(FV_REQUIRE(R && nbls >= 1 && bls, "bad array"); h->impl->set_array(R, nbls, bls, is_coplanar));
}
int fv_sim_set_array_type1(fv_sim *h, const double *basis_matrix, int64_t nbls, const int *bls_int,
                           int n_modes) {
    FV_SIM_CALL(FV_REQUIRE(basis_matrix && nbls >= 1 && bls_int, "bad lattice array"); h->impl->set_array_type1(basis_matrix, nbls, bls_int, n_modes));
}
int fv_sim_set_nbeams(fv_sim *h, int nbeams) {
    FV_SIM_CALL(FV_REQUIRE(nbeams >= 1, "need at least one beam"); h->impl->set_nbeams(nbeams));
}
int fv_sim_set_beam_airy(fv_sim *h, int beam, double diameter) {
    FV_SIM_CALL(FV_REQUIRE(diameter > 0, "diameter must be positive"); h->impl->set_beam_airy(beam, diameter, nullptr, 1.0));
}
int fv_sim_set_reference_compat(fv_sim *h, int on) { FV_SIM_CALL(h->impl->set_reference_compat(on)); }
int fv_sim_set_beam_airy_scaled(fv_sim *h, int beam, double diameter, const double *jones_scale, double power_scale) {
    FV_SIM_CALL(FV_REQUIRE(diameter > 0, "diameter must be positive");
                FV_REQUIRE(power_scale == power_scale, "NaN power factor");
                h->impl->set_beam_airy(beam, diameter, jones_scale, power_scale));
}
int fv_sim_set_beam_table(fv_sim *h, int beam, int nfreq_tab, int nza, int naz, double za_max,
                          const void *table, int order) {
    FV_SIM_CALL(FV_REQUIRE(table, "null table"); h->impl->set_beam_table(beam, nfreq_tab, nza, naz, za_max, table, order));
}
int fv_sim_set_beam_pairs(fv_sim *h, int npairs, const int *bi, const int *bj, const int64_t *off,
                          const int *idx, const signed char *flipped) {
    FV_SIM_CALL(FV_REQUIRE(npairs >= 1 && bi && bj && off, "bad pairs"); h->impl->set_beam_pairs(npairs, bi, bj, off, idx, flipped));
}
int fv_sim_set_basis(fv_sim *h, int nant, int nbasis, int nfreq, const void *coefs, const int *ant1,
                     const int *ant2) {
    FV_SIM_CALL(FV_REQUIRE(nant >= 1 && nbasis >= 1 && coefs && ant1 && ant2, "bad basis"); h->impl->set_basis(nant, nbasis, nfreq, coefs, ant1, ant2));
}
int fv_sim_set_chunking(fv_sim *h, int nchunks, double source_buffer) {
    FV_SIM_CALL(h->impl->set_chunking(nchunks, source_buffer));
}
int fv_sim_run(fv_sim *h, int t0, int t1, int f0, int f1, void *out, int out_on_device) {
    FV_SIM_CALL(FV_REQUIRE(out, "null output"); h->impl->run(t0, t1, f0, f1, out, out_on_device));
}
int fv_sim_run_into(fv_sim *h, int t0, int t1, int f0, int f1, void *out, int64_t out_f_stride, int shared) {
    return guarded([&] {
        FV_REQUIRE(h && h->impl, "null handle");
        FV_REQUIRE(out, "null output");
        FV_REQUIRE(out_f_stride >= 0, "negative channel stride");
        h->impl->out_f_stride = out_f_stride;
        h->impl->out_shared = shared;
        try {
            h->impl->run(t0, t1, f0, f1, out, 0);
        } catch (...) {  // the layout belongs to this call only
            h->impl->out_f_stride = 0;
            h->impl->out_shared = 0;
            throw;
        }
    });
}
int fv_sim_sync(fv_sim *h) { FV_SIM_CALL(h->impl->sync()); }
int fv_sim_stats(fv_sim *h, double *vals, int n) { FV_SIM_CALL(h->impl->stats(vals, n)); }
int fv_sim_reset_stats(fv_sim *h) { FV_SIM_CALL(h->impl->reset_stats()); }
int fv_sim_enable_timing(fv_sim *h, int on) { FV_SIM_CALL(h->impl->enable_timing(on)); }
int fv_sim_timing(fv_sim *h, double *ms, int n) { FV_SIM_CALL(h->impl->timing(ms, n)); }

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Catalog exchange over RCCL for hosts without torch.distributed (SURVEY 8 b / 8 e: one broadcast of the catalog
// from rank 0 at start, optionally only the frequency columns a rank's block needs; there is no other collective on
// the path -- every rank copies its own block of visibilities out).  librccl is opened on first use: the library
// loads, and everything else in it works, on a box without RCCL.
// ---------------------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace fv {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    static Rccl &get() {
        static Rccl *r = [] {
            Rccl *o = new Rccl();  // never destroyed (no static destructor may call into a runtime that is gone)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                o->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (o->lib) break;
            }
            if (!o->lib) return o;
            auto sym = [&](const char *n) { return dlsym(o->lib, n); };
            o->GetUniqueId = reinterpret_cast<decltype(o->GetUniqueId)>(sym("ncclGetUniqueId"));
            o->CommInitRank = reinterpret_cast<decltype(o->CommInitRank)>(sym("ncclCommInitRank"));
            o->CommDestroy = reinterpret_cast<decltype(o->CommDestroy)>(sym("ncclCommDestroy"));
            o->Broadcast = reinterpret_cast<decltype(o->Broadcast)>(sym("ncclBroadcast"));
            o->Send = reinterpret_cast<decltype(o->Send)>(sym("ncclSend"));
            o->Recv = reinterpret_cast<decltype(o->Recv)>(sym("ncclRecv"));
            o->GroupStart = reinterpret_cast<decltype(o->GroupStart)>(sym("ncclGroupStart"));
            o->GroupEnd = reinterpret_cast<decltype(o->GroupEnd)>(sym("ncclGroupEnd"));
            o->GetErrorString = reinterpret_cast<decltype(o->GetErrorString)>(sym("ncclGetErrorString"));
            return o;
        }();
        if (!r->lib || !r->GetUniqueId || !r->CommInitRank || !r->CommDestroy || !r->Broadcast || !r->Send || !r->Recv ||
            !r->GroupStart || !r->GroupEnd)
            throw Error(FV_ERR_INTERNAL, "librccl.so.1 could not be opened (or lacks the nccl* entry points): no RCCL on this host");
        return *r;
    }
    void check(ncclResult_t rc, const char *what) const {
        if (rc == ncclSuccess) return;
        throw Error(FV_ERR_INTERNAL, std::string(what) + ": " + (GetErrorString ? GetErrorString(rc) : "RCCL error") + " (" +
                                         std::to_string((int)rc) + ")");
    }
};

// flux (nsrc, nfreq) of elem_words 4-byte words per entry -> the columns [f0, f1) of every source, contiguous
__global__ void k_pack_columns(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t nsrc, int nfreq,
                               int elem_words, int f0, int f1) {
    const int64_t row_words = (int64_t)(f1 - f0) * elem_words;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsrc * row_words) return;
    const int64_t s = i / row_words, r = i % row_words;
    dst[i] = src[(s * nfreq + f0) * elem_words + r];
}
}  // namespace fv

struct fv_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, nranks = 1;
    hipStream_t stream = nullptr;
    fv::DevBuf staging;
};

extern "C" {

int fv_comm_unique_id(void *id_bytes) {
    return fv::guarded([&] {
        FV_REQUIRE(id_bytes, "null id buffer");
        static_assert(sizeof(ncclUniqueId) == FV_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
        ncclUniqueId id;
        fv::Rccl &r = fv::Rccl::get();
        r.check(r.GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(id_bytes, &id, sizeof(id));
    });
}

int fv_comm_init(fv_comm **c, int device, int rank, int nranks, const void *id_bytes) {
    return fv::guarded([&] {
        FV_REQUIRE(c && id_bytes, "null argument");
        FV_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank outside [0, nranks)");
        *c = nullptr;
        fv::Rccl &r = fv::Rccl::get();
        FV_HIP(hipSetDevice(device));
        std::unique_ptr<fv_comm> o(new fv_comm());
        o->device = device;
        o->rank = rank;
        o->nranks = nranks;
        ncclUniqueId id;
        std::memcpy(&id, id_bytes, sizeof(id));
        r.check(r.CommInitRank(&o->comm, nranks, id, rank), "ncclCommInitRank");
        FV_HIP(hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking));
        *c = o.release();
    });
}

int fv_comm_destroy(fv_comm *c) {
    return fv::guarded([&] {
        if (!c) return;
        (void)hipSetDevice(c->device);
        if (c->stream) {
            (void)hipStreamSynchronize(c->stream);
            (void)hipStreamDestroy(c->stream);
        }
        if (c->comm) (void)fv::Rccl::get().CommDestroy(c->comm);
        delete c;
    });
}

int fv_bcast_catalog(fv_comm *c, int root, void *eq_dev, int64_t eq_bytes, void *flux_dev, int64_t flux_bytes) {
    return fv::guarded([&] {
        FV_REQUIRE(c && c->comm, "null communicator");
        FV_REQUIRE(root >= 0 && root < c->nranks, "root outside [0, nranks)");
        FV_REQUIRE(eq_bytes >= 0 && flux_bytes >= 0 && (eq_dev || !eq_bytes) && (flux_dev || !flux_bytes), "bad buffers");
        fv::Rccl &r = fv::Rccl::get();
        FV_HIP(hipSetDevice(c->device));
        if (eq_bytes) r.check(r.Broadcast(eq_dev, eq_dev, (size_t)eq_bytes, ncclChar, root, c->comm, c->stream), "ncclBroadcast (positions)");
        if (flux_bytes) r.check(r.Broadcast(flux_dev, flux_dev, (size_t)flux_bytes, ncclChar, root, c->comm, c->stream), "ncclBroadcast (flux)");
        FV_HIP(hipStreamSynchronize(c->stream));
    });
}

int fv_scatter_flux_columns(fv_comm *c, int root, int64_t nsrc, int nfreq, int elem_bytes, const void *flux_root_dev,
                            const int *ranges, void *out_dev) {
    return fv::guarded([&] {
        FV_REQUIRE(c && c->comm, "null communicator");
        FV_REQUIRE(root >= 0 && root < c->nranks, "root outside [0, nranks)");
        FV_REQUIRE(nsrc >= 0 && nfreq >= 1 && elem_bytes >= 4 && elem_bytes % 4 == 0, "bad catalog shape");
        FV_REQUIRE(ranges, "null column ranges");
        for (int q = 0; q < c->nranks; ++q)
            FV_REQUIRE(ranges[2 * q] >= 0 && ranges[2 * q] <= ranges[2 * q + 1] && ranges[2 * q + 1] <= nfreq, "column range outside [0, nfreq]");
        const int64_t mine = (int64_t)(ranges[2 * c->rank + 1] - ranges[2 * c->rank]) * nsrc * elem_bytes;
        FV_REQUIRE(out_dev || !mine, "null output");
        FV_REQUIRE(c->rank != root || flux_root_dev || !nsrc, "the root holds the catalog");
        fv::Rccl &r = fv::Rccl::get();
        FV_HIP(hipSetDevice(c->device));
        const int ew = elem_bytes / 4;
        std::vector<int64_t> off(c->nranks + 1, 0);
        if (c->rank == root) {  // pack every rank's columns (one contiguous piece each), then send them in one group
            for (int q = 0; q < c->nranks; ++q) off[q + 1] = off[q] + (int64_t)(ranges[2 * q + 1] - ranges[2 * q]) * nsrc * elem_bytes;
            c->staging.reserve(std::max<size_t>((size_t)off[c->nranks], 16));
            for (int q = 0; q < c->nranks; ++q) {
                const int64_t words = (off[q + 1] - off[q]) / 4;
                if (!words) continue;
                hipLaunchKernelGGL(fv::k_pack_columns, dim3((unsigned)fv::cdiv(words, 256)), dim3(256), 0, c->stream,
                                   static_cast<const uint32_t *>(flux_root_dev),
                                   reinterpret_cast<uint32_t *>(static_cast<char *>(c->staging.p) + off[q]), nsrc, nfreq, ew,
                                   ranges[2 * q], ranges[2 * q + 1]);
            }
            FV_HIP(hipGetLastError());
        }
        r.check(r.GroupStart(), "ncclGroupStart");
        if (c->rank == root)
            for (int q = 0; q < c->nranks; ++q)
                if (off[q + 1] > off[q])
                    r.check(r.Send(static_cast<char *>(c->staging.p) + off[q], (size_t)(off[q + 1] - off[q]), ncclChar, q, c->comm, c->stream), "ncclSend");
        if (mine) r.check(r.Recv(out_dev, (size_t)mine, ncclChar, root, c->comm, c->stream), "ncclRecv");
        r.check(r.GroupEnd(), "ncclGroupEnd");
        FV_HIP(hipStreamSynchronize(c->stream));
    });
}

}  // extern "C"

#ifdef FV_FFT_STAMPS
// diagnostic builds only (see fv_nufft.h): copies out up to max_waves records of 10 values and resets the counter
extern "C" int fv_debug_stamps(unsigned long long *out, int max_waves) {
    unsigned int n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(fv::fv_stamp_count), sizeof(n)) != hipSuccess) return -1;
    if (n > (1u << 18)) n = 1u << 18;
    if ((int)n > max_waves) n = (unsigned)max_waves;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(fv::fv_stamps), sizeof(unsigned long long) * 10 * (size_t)n) != hipSuccess) return -1;
    const unsigned int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(fv::fv_stamp_count), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (int)n;
}
#endif
