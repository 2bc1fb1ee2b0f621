// fv_common.h -- shared host/device helpers for libfftvis_hip (gfx950 only).
#pragma once

#include "../../include/fftvis_hip.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace fv {

// Status codes returned across the C ABI are the FV_* macros of include/fftvis_hip.h.

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define FV_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            throw fv::Error(FV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define FV_REQUIRE(cond, msg)                                        \
    do {                                                             \
        if (!(cond)) throw fv::Error(FV_ERR_ARG, std::string(msg)); \
    } while (0)

// Growable device buffer owned by a plan / engine (never shrinks; freed in dtor).
// bytes of device memory this process holds in DevBufs (all handles): fv_device_bytes()
inline std::atomic<size_t> &dev_bytes_held() {
    static std::atomic<size_t> v{0};
    return v;
}
// ... and the share of it on one device (fv_device_bytes_on): what a run on THAT device can count on reusing
constexpr int DEV_SLOTS = 64;
inline std::atomic<size_t> &dev_bytes_on(int device) {
    static std::atomic<size_t> v[DEV_SLOTS + 1];
    return v[device >= 0 && device < DEV_SLOTS ? device : DEV_SLOTS];
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int dev = -1;  // the device the memory sits on (the current device of the allocating call)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() {
        if (p) (void)hipFree(p);
        dev_bytes_held() -= cap;
        if (cap) dev_bytes_on(dev) -= cap;
    }
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) FV_HIP(hipFree(p));
        dev_bytes_held() -= cap;
        if (cap) dev_bytes_on(dev) -= cap;
        p = nullptr;
        cap = 0;
        FV_HIP(hipMalloc(&p, bytes));
        (void)hipGetDevice(&dev);
        cap = bytes;
        dev_bytes_held() += cap;
        dev_bytes_on(dev) += cap;
    }
    template <typename U>
    U *as() const {
        return reinterpret_cast<U *>(p);
    }
};

template <typename T>
struct cplx {
    T re, im;
};

template <typename T>
__host__ __device__ inline cplx<T> cmul(cplx<T> a, cplx<T> b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename T>
__host__ __device__ inline cplx<T> cconj(cplx<T> a) {
    return {a.re, -a.im};
}
template <typename T>
__host__ __device__ inline cplx<T> cadd(cplx<T> a, cplx<T> b) {
    return {a.re + b.re, a.im + b.im};
}
template <typename T>
__host__ __device__ inline cplx<T> cscale(cplx<T> a, T s) {
    return {a.re * s, a.im * s};
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// roctx ranges around the launches of each kernel family ("prep", "strengths", "spread", "fft", "gather"), so
// that a rocprofv3 --marker-trace / --kernel-trace run groups the kernels by family without parsing names.
// Off unless FFTVIS_HIP_ROCTX=1: the library is looked up at run time (librocprofiler-sdk-roctx, else
// libroctx64), nothing is linked.
struct Roctx {
    using push_t = int (*)(const char *);
    using pop_t = int (*)();
    push_t push = nullptr;
    pop_t pop = nullptr;
    Roctx();
    static Roctx &get() {
        static Roctx r;
        return r;
    }
};
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) : on(Roctx::get().push != nullptr) {
        if (on) Roctx::get().push(name);
    }
    ~RoctxRange() {
        if (on) Roctx::get().pop();
    }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};

// Smallest even integer >= n whose only prime factors are 2, 3, 5 (the usual FFT-friendly sizes).
inline int next235even(int n) {
    if (n <= 2) return 2;
    if (n % 2) ++n;
    for (;; n += 2) {
        int m = n;
        while (m % 2 == 0) m /= 2;
        while (m % 3 == 0) m /= 3;
        while (m % 5 == 0) m /= 5;
        if (m == 1) return n;
    }
}

inline Roctx::Roctx() {
    const char *e = std::getenv("FFTVIS_HIP_ROCTX");
    if (!e || !std::atoi(e)) return;
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
        if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
            push = reinterpret_cast<push_t>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<pop_t>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
}

}  // namespace fv
