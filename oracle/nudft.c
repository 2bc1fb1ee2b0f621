/* oracle/nudft.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Exact direct type-3 non-uniform DFT in fp64:
 *     out[t][k] = sum_j c[t][j] * exp(isign * i * (s_k x_j + t_k y_j + u_k z_j))
 * This is the quantity finufft.nufft2d3 / nufft3d3 approximate to `eps` at the
 * reference's call sites src/fftvis/cpu/nufft.py:48-59 and :105-118 (type 3,
 * isign left at finufft's default +1).  PARITY UNPINNED against finufft itself
 * (package absent from this pipeline) -- see oracle/fftvis_oracle.py header.
 *
 * Built by oracle/Makefile into oracle/libnudft.so; loaded with ctypes by
 * oracle/nudft.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int nudft_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* c and out are interleaved (re,im) row-major (ntrans, M) / (ntrans, N).
 * y/z (and t/u) may be NULL for dim < 2 / dim < 3. */
void nudft_type3_f64(int dim, int64_t M, const double *x, const double *y, const double *z,
                     const double *c, int ntrans, int64_t N, const double *s, const double *t,
                     const double *u, int isign, double *out) {
    const double sg = isign >= 0 ? 1.0 : -1.0;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < N; ++k) {
        const double sk = s[k];
        const double tk = dim > 1 ? t[k] : 0.0;
        const double uk = dim > 2 ? u[k] : 0.0;
        double accr[16], acci[16]; /* ntrans <= 16 per pass */
        for (int t0 = 0; t0 < ntrans; t0 += 16) {
            const int nt = ntrans - t0 < 16 ? ntrans - t0 : 16;
            for (int q = 0; q < nt; ++q) accr[q] = acci[q] = 0.0;
            for (int64_t j = 0; j < M; ++j) {
                double ph = sk * x[j];
                if (dim > 1) ph += tk * y[j];
                if (dim > 2) ph += uk * z[j];
                const double cs = cos(ph), sn = sg * sin(ph);
                for (int q = 0; q < nt; ++q) {
                    const double cr = c[2 * ((int64_t)(t0 + q) * M + j)];
                    const double ci = c[2 * ((int64_t)(t0 + q) * M + j) + 1];
                    accr[q] += cr * cs - ci * sn;
                    acci[q] += cr * sn + ci * cs;
                }
            }
            for (int q = 0; q < nt; ++q) {
                out[2 * ((int64_t)(t0 + q) * N + k)] = accr[q];
                out[2 * ((int64_t)(t0 + q) * N + k) + 1] = acci[q];
            }
        }
    }
}
