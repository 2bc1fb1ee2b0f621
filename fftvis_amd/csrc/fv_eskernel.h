// fv_eskernel.h -- "exponential of semicircle" spreading kernel and its Fourier transform.
//
// The window and its parameter rule follow the published FINUFFT algorithm
// (Barnett, Magland & af Klinteberg, SIAM J. Sci. Comput. 41(5), 2019, sec. 3-4):
//   phi(z) = exp(beta (sqrt(1 - z^2) - 1)),  |z| <= 1,   psi(xi) = phi(2 xi / w),
//   w    = ceil(log10(10/eps))                              (sigma = 2)
//        = ceil(-ln(eps) / (pi sqrt(1 - 1/sigma)))          (other sigma)
//   beta = 2.30 w (sigma = 2; 2.20/2.26/2.38 for w = 2/3/4), else 0.97 pi (1 - 1/(2 sigma)) w.
// The reference reaches this code through finufft.nufft2d3/3d3
// (src/fftvis/cpu/nufft.py:48-59,105-118); nothing here is copied from finufft --
// the evaluation, the quadrature and the data layout are this library's own.
#pragma once

#include "fv_common.h"

namespace fv {

constexpr int MAX_W = 16;   // widest kernel (eps ~ 1e-15 at sigma = 2)
constexpr int MAX_GL = 40;  // Gauss-Legendre nodes on (0, 1) for the kernel's Fourier transform

// Passed by value to kernels (~700 B of kernarg).
struct KerParams {
    int w;        // support in grid cells
    int nq;       // quadrature nodes
    double beta;  // ES shape
    double c;     // 4 / w^2
    double glz[MAX_GL];  // (w/2) z_q            : psi_hat(theta) = sum_q glf[q] cos(theta glz[q])
    double glf[MAX_GL];  // w omega_q phi(z_q)
};

template <typename T>
__host__ __device__ inline T es_eval(T z, T beta, T c) {
    T t = T(1) - c * z * z;
    return t > T(0) ? exp(beta * (sqrt(t) - T(1))) : T(0);
}

__host__ __device__ inline double es_hat(const KerParams &k, double theta) {
    double s = 0.0;
    for (int q = 0; q < k.nq; ++q) s += k.glf[q] * cos(theta * k.glz[q]);
    return s;
}

// Gauss-Legendre nodes/weights on [-1, 1] (Newton on P_n), ascending.
inline void leggauss(int n, std::vector<double> &x, std::vector<double> &wgt) {
    x.assign(n, 0.0);
    wgt.assign(n, 0.0);
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = cos(M_PI * (i + 0.75) / (n + 0.5));
        double pp = 0.0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; ++j) {
                double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            double dz = p1 / pp;
            z -= dz;
            if (fabs(dz) < 1e-16) break;
        }
        x[i] = -z;
        x[n - 1 - i] = z;
        wgt[i] = wgt[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
}

inline KerParams make_kernel(double eps, double sigma, int w_override = 0) {
    KerParams k{};
    int w;
    if (sigma == 2.0)
        w = (int)std::ceil(std::log10(10.0 / eps));
    else  // finufft's low-upsampling width + 1: type 3 applies the kernel twice, and seeded engine fuzzing
        // found 10-16 eps with the bare formula (each extra cell buys a factor ~4).  Capped at 15: the
        // kernel's transform falls by ~e^{-w/2} per dimension across the band at sigma = 1.25, so a
        // wider kernel loses more to amplified rounding at band-edge targets than it gains (w = 16
        // measured 2-3e-8 where w = 15 gives 5-9e-9); ~1e-8 is this sigma's fp64 floor, as in finufft
        w = std::min(15, (int)std::ceil(-std::log(eps) / (M_PI * std::sqrt(1.0 - 1.0 / sigma))) + 1);
    if (w_override > 0) w = w_override;
    w = std::max(2, std::min(MAX_W, w));
    double bow = 2.30;
    if (w == 2) bow = 2.20;
    if (w == 3) bow = 2.26;
    if (w == 4) bow = 2.38;
    if (sigma != 2.0) bow = 0.97 * M_PI * (1.0 - 1.0 / (2.0 * sigma));
    k.w = w;
    k.beta = bow * w;
    k.c = 4.0 / ((double)w * w);
    k.nq = std::min(MAX_GL, 4 + 2 * w);
    std::vector<double> z, om;
    leggauss(2 * k.nq, z, om);
    for (int q = 0; q < k.nq; ++q) {
        double zq = z[k.nq + q];  // positive half
        k.glz[q] = 0.5 * w * zq;
        k.glf[q] = w * om[k.nq + q] * std::exp(k.beta * (std::sqrt(1.0 - zq * zq) - 1.0));
    }
    return k;
}

}  // namespace fv
