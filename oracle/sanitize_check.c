/* oracle/sanitize_check.c -- TEST INFRASTRUCTURE.  Runs the oracle's C restatements (exact NUDFT, CPU
 * spread / interp of the type-3 port) on small inputs; built with -fsanitize=address,undefined by
 * `make -C oracle sanitize`.  Exit status 0 = no sanitizer report, results finite. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void nudft_type3_f64(int dim, int64_t M, const double *x, const double *y, const double *z, const double *c,
                     int ntrans, int64_t N, const double *s, const double *t, const double *u, int isign, double *out);
void cn_spread2d(int64_t M, const double *px, const double *py, const double *c, int ntrans, int w, double beta,
                 int n2x, int n2y, double *grid, int nthreads);
void cn_interp2d(int64_t N, const double *ex, const double *ey, int ntrans, int w, double beta, int n2x, int n2y,
                 const double *grid, double *out, int nthreads);

int main(void) {
    enum { M = 37, N = 11, T = 3, W = 9, NX = 40, NY = 36 };
    double *x = malloc(sizeof(double) * M), *y = malloc(sizeof(double) * M), *c = malloc(sizeof(double) * 2 * T * M);
    double *s = malloc(sizeof(double) * N), *t = malloc(sizeof(double) * N), *out = malloc(sizeof(double) * 2 * T * N);
    double *grid = malloc(sizeof(double) * 2 * T * NX * NY), *px = malloc(sizeof(double) * M), *py = malloc(sizeof(double) * M);
    for (int j = 0; j < M; ++j) {
        x[j] = sin(1.0 + j);
        y[j] = cos(2.0 * j);
        px[j] = 0.5 * W + 1.0 + (NX - W - 2.0) * (0.5 + 0.5 * sin(3.0 * j));  /* footprints stay inside */
        py[j] = 0.5 * W + 1.0 + (NY - W - 2.0) * (0.5 + 0.5 * cos(5.0 * j));
        for (int k = 0; k < T; ++k) {
            c[2 * (k * M + j)] = 1.0 / (1 + j + k);
            c[2 * (k * M + j) + 1] = -0.5 * k;
        }
    }
    for (int k = 0; k < N; ++k) {
        s[k] = 3.0 * k - 10.0;
        t[k] = 7.0 - 1.5 * k;
    }
    int bad = 0;
    for (int thr = 1; thr <= 3; ++thr) {
        nudft_type3_f64(2, M, x, y, NULL, c, T, N, s, t, NULL, thr % 2 ? 1 : -1, out);
        for (int i = 0; i < 2 * T * N; ++i) bad += !isfinite(out[i]);
        cn_spread2d(M, px, py, c, T, W, 2.3 * W, NX, NY, grid, thr);
        double ex[N], ey[N];
        for (int k = 0; k < N; ++k) {
            ex[k] = 0.5 * W + 1.0 + (NX - W - 2.0) * k / (double)N;
            ey[k] = 0.5 * W + 1.0 + (NY - W - 2.0) * (N - 1 - k) / (double)N;
        }
        cn_interp2d(N, ex, ey, T, W, 2.3 * W, NX, NY, grid, out, thr);
        for (int i = 0; i < 2 * T * N; ++i) bad += !isfinite(out[i]);
    }
    free(x); free(y); free(c); free(s); free(t); free(out); free(grid); free(px); free(py);
    printf(bad ? "sanitize_check: %d non-finite values\n" : "sanitize_check: ok\n", bad);
    return bad != 0;
}
