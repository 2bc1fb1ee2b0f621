/* oracle/cpu_nufft.c -- TEST INFRASTRUCTURE / CPU BASELINE, not product code.
 *
 * C/OpenMP spreading and interpolation for the CPU port of the published FINUFFT type-3
 * algorithm in oracle/cpu_nufft.py (the FFT in between is scipy.fft on all cores).  It stands
 * where the reference calls finufft.nufft2d3 (src/fftvis/cpu/nufft.py:48-59); finufft itself is
 * not available in this pipeline, so timings of this code are labelled "port".
 * Built by oracle/Makefile into oracle/libcpunufft.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline double es(double z, double w, double beta) {
    double t = 1.0 - (2.0 * z / w) * (2.0 * z / w);
    return t > 0 ? exp(beta * (sqrt(t) - 1.0)) : 0.0;
}

/* grid[t][y][x] += c[t][j] psi(x - px_j) psi(y - py_j); grid is (ntrans, n2y, n2x) complex,
 * zeroed here.  Threads own horizontal stripes, so no atomics are needed. */
void cn_spread2d(int64_t M, const double *px, const double *py, const double *c, int ntrans, int w,
                 double beta, int n2x, int n2y, double *grid, int nthreads) {
    memset(grid, 0, sizeof(double) * 2 * (size_t)ntrans * n2x * n2y);
#pragma omp parallel num_threads(nthreads)
    {
        int nth = 1, th = 0;
#ifdef _OPENMP
        nth = omp_get_num_threads();
        th = omp_get_thread_num();
#endif
        const int y0 = (int)((int64_t)n2y * th / nth), y1 = (int)((int64_t)n2y * (th + 1) / nth);
        double kx[16], ky[16];
        for (int64_t j = 0; j < M; ++j) {
            const int iy0 = (int)ceil(py[j] - 0.5 * w);
            if (iy0 + w <= y0 || iy0 >= y1) continue;
            const int ix0 = (int)ceil(px[j] - 0.5 * w);
            for (int k = 0; k < w; ++k) {
                kx[k] = es(ix0 + k - px[j], w, beta);
                ky[k] = es(iy0 + k - py[j], w, beta);
            }
            for (int t = 0; t < ntrans; ++t) {
                const double cr = c[2 * ((int64_t)t * M + j)], ci = c[2 * ((int64_t)t * M + j) + 1];
                double *pl = grid + 2 * (size_t)t * n2x * n2y;
                for (int r = 0; r < w; ++r) {
                    const int yy = iy0 + r;
                    if (yy < y0 || yy >= y1) continue;
                    double *row = pl + 2 * ((size_t)yy * n2x + ix0);
                    const double vr = cr * ky[r], vi = ci * ky[r];
                    for (int k = 0; k < w; ++k) {
                        row[2 * k] += vr * kx[k];
                        row[2 * k + 1] += vi * kx[k];
                    }
                }
            }
        }
    }
}

/* out[t][k] = sum grid[t][y][x] psi(x - ex_k) psi(y - ey_k) (-1)^(x+y) */
void cn_interp2d(int64_t N, const double *ex, const double *ey, int ntrans, int w, double beta,
                 int n2x, int n2y, const double *grid, double *out, int nthreads) {
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t k = 0; k < N; ++k) {
        double kx[16], ky[16];
        const int ix0 = (int)ceil(ex[k] - 0.5 * w), iy0 = (int)ceil(ey[k] - 0.5 * w);
        for (int q = 0; q < w; ++q) {
            kx[q] = es(ix0 + q - ex[k], w, beta) * (((ix0 + q) & 1) ? -1.0 : 1.0);
            ky[q] = es(iy0 + q - ey[k], w, beta) * (((iy0 + q) & 1) ? -1.0 : 1.0);
        }
        for (int t = 0; t < ntrans; ++t) {
            const double *pl = grid + 2 * (size_t)t * n2x * n2y;
            double sr = 0, si = 0;
            for (int r = 0; r < w; ++r) {
                const double *row = pl + 2 * ((size_t)(iy0 + r) * n2x + ix0);
                double tr = 0, ti = 0;
                for (int q = 0; q < w; ++q) {
                    tr += row[2 * q] * kx[q];
                    ti += row[2 * q + 1] * kx[q];
                }
                sr += tr * ky[r];
                si += ti * ky[r];
            }
            out[2 * ((int64_t)t * N + k)] = sr;
            out[2 * ((int64_t)t * N + k) + 1] = si;
        }
    }
}
