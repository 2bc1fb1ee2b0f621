// fv_sim.h -- the fused visibility simulator: per time  rotate -> horizon compaction -> az/za;
// per frequency group  beam x coherency strengths -> type-3 NUFFT (spread, pruned row FFTs, gather).
//
// GPU twin of CPUSimulationEngine._evaluate_vis_chunk (src/fftvis/cpu/cpu_simulate.py:856-1071).
// Loop order follows the reference (time -> frequency -> beam pair); what differs is that a
// block of neighbouring frequencies shares one fine grid geometry, so their strengths ride one
// spread launch and one batched FFT as extra "transforms" (DESIGN.md "Frequency groups").
#pragma once

#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include "fv_nufft.h"
#include <unordered_map>

#include <algorithm>
#include <cstdlib>
#include <memory>

namespace fv {

constexpr double SPEED_OF_LIGHT = 299792458.0;  // core/utils.py:9

struct BeamDesc {
    int kind;            // 0 Airy, 1 table
    int order;           // tables: interpolation order 0..5 (read by the general-order path only, eval_* <ORD = 0>)
    double diameter;     // Airy
    double js[8];        // Airy: complex factor per Jones slot A[ax][feed] = js . 2 J1(x)/x  (re, im pairs)
    double ps;           // Airy: factor of the power beam, ps . (2 J1(x)/x)^2
    const void *table;   // device
    int nfreq_tab, nza, naz;
    double za_max;
};

// ---------------------------------------------------------------------------------------------
// per-time kernels
// ---------------------------------------------------------------------------------------------
struct Rot9 {
    double m[9];
};

// Pass 1: above-horizon flag count per 256-source block (select_chunk's up > 0,
// cpu_simulate.py:940-946 via matvis).
// The kernels work on the source range [off, off + n) of the catalog (eq is (3, stride) SoA): one
// source chunk of the reference's `for chunk in range(nchunks)` loop (cpu_simulate.py:939).
template <typename T>
__global__ void k_horizon_count(int64_t n, int64_t stride, int64_t off, const T *__restrict__ eq, Rot9 rt,
                                int *__restrict__ block_counts) {
    __shared__ int wsum[4];
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool up = false;
    if (j < n) {
        double ex = eq[off + j], ey = eq[stride + off + j], ez = eq[2 * stride + off + j];
        up = rt.m[6] * ex + rt.m[7] * ey + rt.m[8] * ez > 0.0;
    }
    unsigned long long b = __ballot(up);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Pass 2: stable compaction + everything that depends only on (time, source):
//   topo = R_t eq;  az, za in the UN-rotated ENU frame (cpu_simulate.py:957-959, matvis
//   enu_to_az_za "uvbeam");  x = 2 pi R_plane topo (cpu_simulate.py:961-967).
// cap = capacity of the compacted arrays (chunk size x source_buffer): sources beyond it are not
// stored and counted in *overflow (matvis raises likewise when its above-horizon buffer is too small).
template <typename T>
__global__ void k_horizon_compact(int64_t nsrc, int64_t stride, int64_t off, const T *__restrict__ eq, Rot9 rt,
                                  Rot9 rp, const int *__restrict__ block_off, T *__restrict__ xyz,
                                  int64_t cap, T *__restrict__ az, T *__restrict__ za,
                                  int *__restrict__ src_idx, int *__restrict__ overflow) {
    __shared__ int wsum[4];
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double e = 0, n = 0, u = -1;
    if (j < nsrc) {
        double ex = eq[off + j], ey = eq[stride + off + j], ez = eq[2 * stride + off + j];
        e = rt.m[0] * ex + rt.m[1] * ey + rt.m[2] * ez;
        n = rt.m[3] * ex + rt.m[4] * ey + rt.m[5] * ez;
        u = rt.m[6] * ex + rt.m[7] * ey + rt.m[8] * ez;
    }
    const bool up = j < nsrc && u > 0.0;
    unsigned long long b = __ballot(up);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = __popcll(b);
    __syncthreads();
    int base = block_off[blockIdx.x];
    for (int i = 0; i < wv; ++i) base += wsum[i];
    if (!up) return;
    const int64_t pos = base + __popcll(b & ((1ull << lane) - 1ull));
    if (pos >= cap) {
        atomicAdd(overflow, 1);
        return;
    }
    const double lsqr = n * n + e * e;
    const double zeta = sqrt(fmax(0.0, 1.0 - lsqr));
    double azv = 0.5 * M_PI - atan2(e, n);
    azv = fmod(azv, 2.0 * M_PI);
    if (azv < 0) azv += 2.0 * M_PI;
    az[pos] = (T)azv;
    za[pos] = (T)(0.5 * M_PI - asin(zeta));
    const double twopi = 2.0 * M_PI;
    xyz[pos] = (T)(twopi * (rp.m[0] * e + rp.m[1] * n + rp.m[2] * u));
    xyz[cap + pos] = (T)(twopi * (rp.m[3] * e + rp.m[4] * n + rp.m[5] * u));
    xyz[2 * cap + pos] = (T)(twopi * (rp.m[6] * e + rp.m[7] * n + rp.m[8] * u));
    src_idx[pos] = (int)(off + j);
}

// ---------------------------------------------------------------------------------------------
// beam x coherency  (evaluate_beam cpu/beams.py:12-89; _compute_apparent_coherency
// cpu_simulate.py:90-202; numba kernels cpu/beams.py:129-246)
// ---------------------------------------------------------------------------------------------
__device__ inline double airy_efield(double diameter, double freq, double za) {
    const double x = M_PI * diameter * freq * sin(za) / SPEED_OF_LIGHT;
    return x == 0.0 ? 1.0 : 2.0 * j1(x) / x;
}

struct Bilin {
    int ia0, ia1, iz0, iz1;
    double wa, wz;
};

__device__ inline Bilin bilin_setup(const BeamDesc &b, double az, double za) {
    Bilin o;
    const double twopi = 2.0 * M_PI;
    double a = fmod(az, twopi);
    if (a < 0) a += twopi;
    const double fa = a / (twopi / b.naz);
    int ia0 = (int)floor(fa);
    o.wa = fa - ia0;
    ia0 %= b.naz;
    o.ia0 = ia0;
    o.ia1 = (ia0 + 1) % b.naz;
    double fz = za / (b.za_max / (b.nza - 1));
    fz = fmin(fmax(fz, 0.0), (double)(b.nza - 1));
    int iz0 = min((int)floor(fz), b.nza - 2);
    o.wz = fz - iz0;
    o.iz0 = iz0;
    o.iz1 = iz0 + 1;
    return o;
}

// Order-3 interpolation (beam_spline_opts {"order": 3}, the reference CLI's default, cli.py:50,146;
// scipy.ndimage.map_coordinates semantics: interpolating cubic B-spline): nodes floor(x) - 1 .. + 2
// with the B-spline weights, on a table whose samples were replaced by spline coefficients at upload
// (k_bspline3_prefilter).  az is periodic; za mirrors about its first and last node.
struct Cubic {
    int ia[4], iz[4];
    double wa[4], wz[4];
};

__device__ inline void bspline3_weights(double t, double w[4]) {
    const double u = 1.0 - t, t2 = t * t, t3 = t2 * t;
    w[0] = u * u * u * (1.0 / 6.0);
    w[1] = (4.0 - 6.0 * t2 + 3.0 * t3) * (1.0 / 6.0);
    w[2] = (1.0 + 3.0 * t + 3.0 * t2 - 3.0 * t3) * (1.0 / 6.0);
    w[3] = t3 * (1.0 / 6.0);
}

__device__ inline Cubic cubic_setup(const BeamDesc &b, double az, double za) {
    Cubic o;
    const double twopi = 2.0 * M_PI;
    double a = fmod(az, twopi);
    if (a < 0) a += twopi;
    const double fa = a / (twopi / b.naz);
    const int ia0 = (int)floor(fa);
    bspline3_weights(fa - ia0, o.wa);
    for (int k = 0; k < 4; ++k) o.ia[k] = ((ia0 - 1 + k) % b.naz + b.naz) % b.naz;
    double fz = za / (b.za_max / (b.nza - 1));
    fz = fmin(fmax(fz, 0.0), (double)(b.nza - 1));
    const int iz0 = min((int)floor(fz), b.nza - 2);
    bspline3_weights(fz - iz0, o.wz);
    const int per = 2 * (b.nza - 1);
    for (int k = 0; k < 4; ++k) {
        int j = ((iz0 - 1 + k) % per + per) % per;
        o.iz[k] = j < b.nza ? j : per - j;
    }
    return o;
}

// Any order 0 .. 5 (beam_spline_opts {"order": n}; scipy.ndimage.map_coordinates semantics as for order 3): the
// n + 1 nodes start at floor(x) - n / 2 (odd n) or floor(x + 1/2) - n / 2 (even n: the centred B-spline's knots sit
// at half-integers) and carry the cardinal B-spline's values, built by the Cox - de Boor triangle on uniform knots
// (every denominator is the level j).  Orders 1 and 3 have their own unrolled paths above; this one runs the
// others (0: nearest node; 2, 4, 5: on coefficients from k_bspline_prefilter) with the order read at run time.
struct SplineN {
    int n;  // order
    int ia[6], iz[6];
    double wa[6], wz[6];
};

__device__ inline void bspline_weights(int n, double t, double w[6]) {  // t in [0, 1]: position inside the knot span
    w[0] = 1.0;
    for (int j = 1; j <= n; ++j) {
        double saved = 0.0;
        const double inv = 1.0 / j;
        for (int r = 0; r < j; ++r) {
            const double right = r + 1 - t, left = t + (j - r - 1);
            const double tmp = w[r] * inv;
            w[r] = saved + right * tmp;
            saved = left * tmp;
        }
        w[j] = saved;
    }
}

__device__ inline SplineN spline_setup(const BeamDesc &b, double az, double za) {
    SplineN o;
    const int n = o.n = b.order;
    const double half = n & 1 ? 0.0 : 0.5;
    const double twopi = 2.0 * M_PI;
    double a = fmod(az, twopi);
    if (a < 0) a += twopi;
    const double fa = a / (twopi / b.naz) + half;
    const int ia0 = (int)floor(fa);
    bspline_weights(n, fa - ia0, o.wa);
    for (int k = 0; k <= n; ++k) o.ia[k] = ((ia0 - n / 2 + k) % b.naz + b.naz) % b.naz;
    double fz = za / (b.za_max / (b.nza - 1));
    fz = fmin(fmax(fz, 0.0), (double)(b.nza - 1)) + half;
    const int iz0 = n & 1 ? min((int)floor(fz), b.nza - 2) : (int)floor(fz);
    bspline_weights(n, fz - iz0, o.wz);
    const int per = 2 * (b.nza - 1);
    for (int k = 0; k <= n; ++k) {
        const int j = ((iz0 - n / 2 + k) % per + per) % per;
        o.iz[k] = j < b.nza ? j : per - j;
    }
    return o;
}

// scipy.ndimage.spline_filter1d(order) in place, one line per thread, on a table stored as
// [freq][za][az][C] doubles: axis 0 = za lines (mode "mirror"), axis 1 = az lines ("grid-wrap").
// One causal + anticausal sweep per pole (order 2: sqrt(8) - 3; 3: sqrt(3) - 2; 4 and 5: two poles each), after the
// gain prod (1 - z)(1 - 1 / z); boundary sums run over the whole line (exact, as scipy's).
__global__ void k_bspline_prefilter(double *__restrict__ data, int64_t nfreq, int nza, int naz, int C,
                                    int axis, int order) {
    const int nother = axis == 0 ? naz : nza;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nfreq * nother * C) return;
    const int c = (int)(t % C);
    const int64_t r = t / C;
    const int other = (int)(r % nother);
    const int64_t f = r / nother;
    double *p;
    int64_t st;
    int n;
    if (axis == 0) {
        p = data + ((f * nza) * naz + other) * C + c;
        st = (int64_t)naz * C;
        n = nza;
    } else {
        p = data + ((f * nza + other) * naz) * C + c;
        st = C;
        n = naz;
    }
    if (n < 2) return;
    double poles[2];
    int npoles = 1;
    switch (order) {
        case 2: poles[0] = sqrt(8.0) - 3.0; break;
        case 3: poles[0] = sqrt(3.0) - 2.0; break;
        case 4:
            npoles = 2;
            poles[0] = sqrt(664.0 - sqrt(438976.0)) + sqrt(304.0) - 19.0;
            poles[1] = sqrt(664.0 + sqrt(438976.0)) - sqrt(304.0) - 19.0;
            break;
        case 5:
            npoles = 2;
            poles[0] = sqrt(67.5 - sqrt(4436.25)) + sqrt(26.25) - 6.5;
            poles[1] = sqrt(67.5 + sqrt(4436.25)) - sqrt(26.25) - 6.5;
            break;
        default: return;  // orders 0 and 1 interpolate the samples themselves
    }
    double gain = 1.0;
    for (int q = 0; q < npoles; ++q) gain *= (1.0 - poles[q]) * (1.0 - 1.0 / poles[q]);
    for (int i = 0; i < n; ++i) p[i * st] *= gain;
    for (int q = 0; q < npoles; ++q) {
    const double z = poles[q];
    if (axis == 0) {  // mirror
        const double zn1 = pow(z, (double)(n - 1));
        double c0 = p[0] + zn1 * p[(n - 1) * st], zi = z;
        for (int i = 1; i < n - 1; ++i) {
            c0 += zi * (p[i * st] + zn1 * p[(n - 1 - i) * st]);
            zi *= z;
        }
        p[0] = c0 / (1.0 - zn1 * zn1);
    } else {  // periodic
        double c0 = p[0], zi = z;
        for (int i = 1; i < n; ++i) {
            c0 += zi * p[(n - i) * st];
            zi *= z;
        }
        p[0] = c0 / (1.0 - zi);
    }
    double prev = p[0];
    for (int i = 1; i < n; ++i) {
        prev = p[i * st] + z * prev;
        p[i * st] = prev;
    }
    if (axis == 0) {
        p[(n - 1) * st] = (z * p[(n - 2) * st] + p[(n - 1) * st]) * z / (z * z - 1.0);
    } else {
        double cl = p[(n - 1) * st], zi = z;
        for (int i = 0; i < n - 1; ++i) {
            cl += zi * p[i * st];
            zi *= z;
        }
        p[(n - 1) * st] = cl * z / (zi - 1.0);
    }
    double next = p[(n - 1) * st];
    for (int i = n - 2; i >= 0; --i) {
        next = z * (next - p[i * st]);
        p[i * st] = next;
    }
    }  // poles
}

// [freq][4][za][az] (caller's layout) -> [freq][za][az][4] (device layout of Jones tables)
__global__ void k_jones_interleave(const cplx<double> *__restrict__ in, cplx<double> *__restrict__ out,
                                   int64_t nodes, int64_t nfreq) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes * nfreq) return;
    const int64_t f = i / nodes, nd = i % nodes;
    for (int j = 0; j < 4; ++j) out[(f * nodes + nd) * 4 + j] = in[(f * 4 + j) * nodes + nd];
}

// samples -> B-spline coefficients of the given order, both axes (orders >= 2; once per upload)
inline void bspline_prefilter(double *table, int64_t nfreq, int nza, int naz, int C, int order, hipStream_t s) {
    if (order < 2) return;
    for (int axis = 0; axis < 2; ++axis) {
        const int64_t lines = nfreq * (axis == 0 ? naz : nza) * C;
        hipLaunchKernelGGL(k_bspline_prefilter, dim3((unsigned)cdiv(lines, 64)), dim3(64), 0, s, table, nfreq,
                           nza, naz, C, axis, order);
    }
}
// kernel variant for an order: 1 and 3 unrolled, everything else through the general path (template value 0)
inline int beam_order_variant(int order) { return order == 1 || order == 3 ? order : 0; }

// Jones matrix A[ax][feed] (row-major, 4 complex) of one beam at one (source, frequency).
template <int ORD>
__device__ inline void eval_jones(const BeamDesc &b, int fidx, double freq, double az, double za,
                                  cplx<double> A[4]) {
    if (b.kind == 0) {
        const double e = airy_efield(b.diameter, freq, za);
        for (int i = 0; i < 4; ++i) A[i] = {e * b.js[2 * i], e * b.js[2 * i + 1]};
        return;
    }
    const int ft = b.nfreq_tab > 1 ? fidx : 0;
    if (ORD == 0) {
        const SplineN w = spline_setup(b, az, za);
        const cplx<double> *tab = (const cplx<double> *)b.table + (int64_t)ft * 4 * b.nza * b.naz;
        for (int i = 0; i < 4; ++i) A[i] = {0.0, 0.0};
        for (int k = 0; k <= w.n; ++k)
            for (int l = 0; l <= w.n; ++l) {
                const double wt = w.wz[k] * w.wa[l];
                const cplx<double> *nd = tab + ((int64_t)w.iz[k] * b.naz + w.ia[l]) * 4;
                for (int i = 0; i < 4; ++i) {
                    const cplx<double> v = nd[i];
                    A[i].re += v.re * wt;
                    A[i].im += v.im * wt;
                }
            }
        return;
    }
    if (ORD == 3) {
        const Cubic w = cubic_setup(b, az, za);
        const cplx<double> *tab = (const cplx<double> *)b.table + (int64_t)ft * 4 * b.nza * b.naz;
        for (int i = 0; i < 4; ++i) A[i] = {0.0, 0.0};
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 4; ++l) {
                const double wt = w.wz[k] * w.wa[l];
                const cplx<double> *nd = tab + ((int64_t)w.iz[k] * b.naz + w.ia[l]) * 4;
                for (int i = 0; i < 4; ++i) {
                    const cplx<double> v = nd[i];
                    A[i].re += v.re * wt;
                    A[i].im += v.im * wt;
                }
            }
        return;
    }
    const Bilin w = bilin_setup(b, az, za);
    // device layout [freq][za][az][4 Jones]: the four Jones entries of a node are one 64-B sector (the
    // caller's [freq][2][2][za][az] planes are interleaved at upload, k_jones_interleave)
    const cplx<double> *tab = (const cplx<double> *)b.table + (int64_t)ft * 4 * b.nza * b.naz;
    const double w00 = (1 - w.wz) * (1 - w.wa), w01 = (1 - w.wz) * w.wa, w10 = w.wz * (1 - w.wa),
                 w11 = w.wz * w.wa;
    const cplx<double> *n00 = tab + ((int64_t)w.iz0 * b.naz + w.ia0) * 4, *n01 = tab + ((int64_t)w.iz0 * b.naz + w.ia1) * 4,
                       *n10 = tab + ((int64_t)w.iz1 * b.naz + w.ia0) * 4, *n11 = tab + ((int64_t)w.iz1 * b.naz + w.ia1) * 4;
    for (int i = 0; i < 4; ++i) {
        const cplx<double> v00 = n00[i], v01 = n01[i], v10 = n10[i], v11 = n11[i];
        A[i] = {v00.re * w00 + v01.re * w01 + v10.re * w10 + v11.re * w11,
                v00.im * w00 + v01.im * w01 + v10.im * w10 + v11.im * w11};
    }
}

// Power beam (unpolarized path: prepare_beam_unpolarized in wrapper.py:278-279 hands the engine a
// single-polarisation power beam; evaluate_beam returns [0,0,0,:], cpu/beams.py:78-81).
template <int ORD>
__device__ inline double eval_power(const BeamDesc &b, int fidx, double freq, double az, double za) {
    if (b.kind == 0) {
        const double e = airy_efield(b.diameter, freq, za);
        return e * e * b.ps;
    }
    const int ft = b.nfreq_tab > 1 ? fidx : 0;
    const double *p = (const double *)b.table + (int64_t)ft * b.nza * b.naz;
    if (ORD == 0) {
        const SplineN w = spline_setup(b, az, za);
        double acc = 0.0;
        for (int k = 0; k <= w.n; ++k)
            for (int l = 0; l <= w.n; ++l) acc += w.wz[k] * w.wa[l] * p[(int64_t)w.iz[k] * b.naz + w.ia[l]];
        return acc;
    }
    if (ORD == 3) {
        const Cubic w = cubic_setup(b, az, za);
        double acc = 0.0;
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 4; ++l) acc += w.wz[k] * w.wa[l] * p[(int64_t)w.iz[k] * b.naz + w.ia[l]];
        return acc;
    }
    const Bilin w = bilin_setup(b, az, za);
    return p[(int64_t)w.iz0 * b.naz + w.ia0] * (1 - w.wz) * (1 - w.wa) +
           p[(int64_t)w.iz0 * b.naz + w.ia1] * (1 - w.wz) * w.wa +
           p[(int64_t)w.iz1 * b.naz + w.ia0] * w.wz * (1 - w.wa) +
           p[(int64_t)w.iz1 * b.naz + w.ia1] * w.wz * w.wa;
}

// out[a][p] = sum_b conj(Ai[b][a]) Aj[b][p] * I      (cpu/beams.py:129-145, 182-212;
// einsum "bas,s,bps->aps" in tests/test_cpu_beams.py:870)
__host__ __device__ inline void coh_AhB_flux(const cplx<double> Ai[4], const cplx<double> Aj[4],
                                             double I, cplx<double> o[4]) {
    for (int a = 0; a < 2; ++a)
        for (int p = 0; p < 2; ++p)
            o[a * 2 + p] = cscale(cadd(cmul(cconj(Ai[a]), Aj[p]), cmul(cconj(Ai[2 + a]), Aj[2 + p])), I);
}

// out[a][p] = sum_{b,k} conj(Ai[b][a]) C[b][k] Aj[k][p]   (cpu/beams.py:147-180, 215-246;
// einsum "bas,bks,kps->aps" in tests/test_cpu_beams.py:953)
__host__ __device__ inline void coh_AhCB(const cplx<double> Ai[4], const cplx<double> C[4],
                                         const cplx<double> Aj[4], cplx<double> o[4]) {
    for (int a = 0; a < 2; ++a) {
        const cplx<double> t0 = cadd(cmul(cconj(Ai[a]), C[0]), cmul(cconj(Ai[2 + a]), C[2]));
        const cplx<double> t1 = cadd(cmul(cconj(Ai[a]), C[1]), cmul(cconj(Ai[2 + a]), C[3]));
        for (int p = 0; p < 2; ++p) o[a * 2 + p] = cadd(cmul(t0, Aj[p]), cmul(t1, Aj[2 + p]));
    }
}

__host__ __device__ inline cplx<double> csqrt_principal(cplx<double> z) {
    const double r = hypot(z.re, z.im);
    if (r == 0.0) return {0.0, 0.0};
    double sr = sqrt(0.5 * (r + fabs(z.re)));
    double si = 0.5 * z.im / sr;
    if (z.re < 0) {  // swap so that the real part stays >= 0
        const double t = sr;
        sr = fabs(si);
        si = z.im < 0 ? -t : t;
    }
    return {sr, si};
}

// Stand-alone coherency op on reference-layout arrays (2, 2, n) [a][b][src]:
// variant 0: beam <- (A^H A) I          get_apparent_flux_polarized_beam   cpu/beams.py:129-145
//         1: beam <- A^H C A            get_apparent_flux_polarized        cpu/beams.py:147-180
//         2: out  <- Ai^H Aj I          ..._polarized_beam_pair            cpu/beams.py:182-212
//         3: out  <- Ai^H C Aj          ..._polarized_pair                 cpu/beams.py:215-246
//         4: out  <- sqrt(Bi Bj) I      unpolarized, (n) arrays            cpu_simulate.py:183-187
template <typename T>
__global__ void k_apparent_coherency(int variant, int64_t n, const cplx<T> *__restrict__ bi,
                                     const cplx<T> *__restrict__ bj, const void *__restrict__ flux,
                                     cplx<T> *__restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    if (variant == 4) {
        const cplx<double> a = {(double)bi[s].re, (double)bi[s].im}, b = {(double)bj[s].re, (double)bj[s].im};
        const cplx<double> r = cscale(csqrt_principal(cmul(a, b)), (double)((const T *)flux)[s]);
        out[s] = {(T)r.re, (T)r.im};
        return;
    }
    cplx<double> Ai[4], Aj[4], o[4];
    for (int i = 0; i < 4; ++i) {
        Ai[i] = {(double)bi[i * n + s].re, (double)bi[i * n + s].im};
        Aj[i] = {(double)bj[i * n + s].re, (double)bj[i * n + s].im};
    }
    if (variant == 0 || variant == 2) {
        coh_AhB_flux(Ai, Aj, (double)((const T *)flux)[s], o);
    } else {
        const cplx<T> *Cp = (const cplx<T> *)flux;
        cplx<double> C[4];
        for (int i = 0; i < 4; ++i) C[i] = {(double)Cp[i * n + s].re, (double)Cp[i * n + s].im};
        coh_AhCB(Ai, C, Aj, o);
    }
    for (int i = 0; i < 4; ++i) out[i * n + s] = {(T)o[i].re, (T)o[i].im};
}

// Stand-alone beam evaluation (GPUBeamEvaluator.evaluate_beam, gpu/beams.py:18-66):
// polarized -> (2, 2, n) [ax][feed][src]; else (n) power (imaginary part 0).
template <typename T, int ORD>
__global__ void k_beam_eval(BeamDesc b, int polarized, int fidx, double freq, int64_t n,
                            const T *__restrict__ az, const T *__restrict__ za,
                            cplx<T> *__restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    if (!polarized) {
        out[s] = {(T)eval_power<ORD>(b, fidx, freq, (double)az[s], (double)za[s]), T(0)};
        return;
    }
    cplx<double> A[4];
    eval_jones<ORD>(b, fidx, freq, (double)az[s], (double)za[s], A);
    for (int i = 0; i < 4; ++i) out[i * n + s] = {(T)A[i].re, (T)A[i].im};
}

// b[:, j] <- rot b[:, j]   (gpu/utils.py:8-22; cpu/utils.py:5-24)
template <typename T>
__global__ void k_inplace_rot(Rot9 r, T *__restrict__ b, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double x = b[j], y = b[n + j], z = b[2 * n + j];
    b[j] = (T)(r.m[0] * x + r.m[1] * y + r.m[2] * z);
    b[n + j] = (T)(r.m[3] * x + r.m[4] * y + r.m[5] * z);
    b[2 * n + j] = (T)(r.m[6] * x + r.m[7] * y + r.m[8] * z);
}

// ---------------------------------------------------------------------------------------------
// per-source astrometry from a per-time context (reference cpu_simulate.py:693-709,937)
// ---------------------------------------------------------------------------------------------
// ICRS -> observed is a source-independent context per time -- what ERFA keeps in eraASTROM and erfa.apco13 (or
// astropy's erfa_astrom) fills in microseconds on the host -- applied to every source: light deflection by the Sun,
// annual aberration, bias-precession-nutation (ICRS -> CIRS: the published algorithm of eraAtciqz = eraLdsun + eraAb +
// the BPN matrix), then Earth rotation, polar motion, diurnal aberration, the rotation to the local horizon and the
// A tan z + B tan^3 z refraction (CIRS -> observed: eraAtioq).  The context is taken in eraASTROM's field order so
// that a caller can hand over the bytes of the array ERFA filled.  One thread per source; arithmetic in fp64 whatever
// the catalog's precision.  Output: topocentric (east, north, up) unit vectors, the layout of fv_sim_set_topo.
// PARITY UNPINNED versus ERFA / matvis (neither is in this pipeline): pinned to the numpy restatement in oracle/.
struct Astrom {
    double pmt, eb[3], eh[3], em, v[3], bm1, bpn[9], along, phi, xpl, ypl, sphi, cphi, diurab, eral, refa, refb;
};
static_assert(sizeof(Astrom) == 31 * sizeof(double), "eraASTROM is 31 doubles");

__device__ inline void astrom_icrs_to_enu(const Astrom &a, double px, double py, double pz, double *enu) {
    constexpr double SRS = 1.97412574336e-8;  // Schwarzschild radius of the Sun in au
    // light deflection by the Sun (unit mass; q = p: the source is at infinity)
    const double em2 = fmax(a.em * a.em, 1.0), dlim = 1e-6 / em2;
    const double qdqpe = px * (px + a.eh[0]) + py * (py + a.eh[1]) + pz * (pz + a.eh[2]);
    const double wd = SRS / a.em / fmax(qdqpe, dlim);
    const double ex = a.eh[1] * pz - a.eh[2] * py, ey = a.eh[2] * px - a.eh[0] * pz, ez = a.eh[0] * py - a.eh[1] * px;  // e x q
    double qx = px + wd * (py * ez - pz * ey), qy = py + wd * (pz * ex - px * ez), qz = pz + wd * (px * ey - py * ex);  // p + w p x (e x q)
    // annual aberration (relativistic, with the Sun's potential term)
    const double pdv = qx * a.v[0] + qy * a.v[1] + qz * a.v[2];
    const double w1 = 1.0 + pdv / (1.0 + a.bm1), w2 = SRS / a.em;
    double ax = qx * a.bm1 + w1 * a.v[0] + w2 * (a.v[0] - pdv * qx);
    double ay = qy * a.bm1 + w1 * a.v[1] + w2 * (a.v[1] - pdv * qy);
    double az = qz * a.bm1 + w1 * a.v[2] + w2 * (a.v[2] - pdv * qz);
    const double rn = 1.0 / sqrt(ax * ax + ay * ay + az * az);
    ax *= rn;
    ay *= rn;
    az *= rn;
    // bias-precession-nutation: CIRS
    const double cx = a.bpn[0] * ax + a.bpn[1] * ay + a.bpn[2] * az;
    const double cy = a.bpn[3] * ax + a.bpn[4] * ay + a.bpn[5] * az;
    const double cz = a.bpn[6] * ax + a.bpn[7] * ay + a.bpn[8] * az;
    // Earth rotation: (-HA, Dec) Cartesian
    double se, ce;
    sincos(a.eral, &se, &ce);
    const double x = ce * cx + se * cy, y = -se * cx + ce * cy, z = cz;
    // polar motion
    double sx, cxp, sy, cyp;
    sincos(a.xpl, &sx, &cxp);
    sincos(a.ypl, &sy, &cyp);
    const double xhd = cxp * x + sx * z;
    const double yhd = sx * sy * x + cyp * y - cxp * sy * z;
    const double zhd = -sx * cyp * x + sy * y + cxp * cyp * z;
    // diurnal aberration
    const double f = 1.0 - a.diurab * yhd;
    const double xhdt = f * xhd, yhdt = f * (yhd + a.diurab), zhdt = f * zhd;
    // to the horizon frame (x south -> north is -x, y east, z up)
    const double xaet = a.sphi * xhdt - a.cphi * zhdt, yaet = yhdt, zaet = a.cphi * xhdt + a.sphi * zhdt;
    // refraction, A tan z + B tan^3 z with ERFA's guards near the horizon (identity for refa = refb = 0)
    double r = sqrt(xaet * xaet + yaet * yaet);
    r = r > 1e-6 ? r : 1e-6;
    const double zc = zaet > 0.05 ? zaet : 0.05;
    const double tz = r / zc, wr = a.refb * tz * tz;
    const double del = (a.refa + wr) * tz / (1.0 + (a.refa + 3.0 * wr) / (zc * zc));
    const double cosdel = 1.0 - del * del / 2.0, fr = cosdel - del * zc / r;
    const double xo = xaet * fr, yo = yaet * fr, zo = cosdel * zaet + del * r;
    const double on = 1.0 / sqrt(xo * xo + yo * yo + zo * zo);
    enu[0] = yo * on;
    enu[1] = -xo * on;
    enu[2] = zo * on;
}

// sources [off, off + n) of the (3, stride) catalog -> the same positions of a (3, stride) array of ENU vectors
template <typename T>
__global__ void k_astrom_topo(int64_t n, int64_t stride, int64_t off, const T *__restrict__ eq, Astrom a,
                              T *__restrict__ topo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t j = off + i;
    double enu[3];
    astrom_icrs_to_enu(a, (double)eq[j], (double)eq[stride + j], (double)eq[2 * stride + j], enu);
    topo[j] = (T)enu[0];
    topo[stride + j] = (T)enu[1];
    topo[2 * stride + j] = (T)enu[2];
}

struct StrengthArgs {
    int64_t M;          // capacity of the per-time arrays (stride); live count is *Mp
    int nfg;            // frequencies in this group
    int f_first;        // catalog index of the group's first frequency
    int nfreq;          // catalog frequency count (flux row length)
    int polarized, pol_sky, same_beam;
    int herm;           // strengths packed as two transforms (see k_interp): 1 Hermitian (c_00 + i c_11, c_01),
                        // 2 all real (c_00 + i c_11, c_01 + i c_10)
    int dim, w;
    double h[3], btc[3];
    int na[3];
    BeamDesc bi, bj;
    // height term k of a nearly flat array (Sim::run): strengths times ((z - wt_zc) wt_inv)^k, z = the sources' third
    // coordinate (compacted order).  A real factor: the Hermitian / all-real packings stay what they are.  wt_k <= 0: none.
    int wt_k;
    double wt_zc, wt_inv;
};

// Strengths of compacted source jc at catalog frequency fidx for one beam pair, times `pre`:
// tpol values written to dst.
template <typename T, int ORD>
__device__ inline void strength_eval(const StrengthArgs &a, int jc, int fidx, cplx<double> pre,
                                     const int *__restrict__ src_idx, const T *__restrict__ az,
                                     const T *__restrict__ za, const void *__restrict__ flux,
                                     const double *__restrict__ freqs, cplx<T> *__restrict__ dst) {
    const double freq = freqs[fidx];
    const int64_t js = src_idx[jc];   // catalog index
    const double azv = az[jc], zav = za[jc];
    if (!a.polarized) {
        // cpu_simulate.py:183-187: sqrt(B_i B_j) * I   (principal square root)
        const double bi = eval_power<ORD>(a.bi, fidx, freq, azv, zav);
        const double bj = a.same_beam ? bi : eval_power<ORD>(a.bj, fidx, freq, azv, zav);
        const double I = (double)((const T *)flux)[js * a.nfreq + fidx];
        cplx<double> c = cscale(csqrt_principal(cplx<double>{bi * bj, 0.0}), I);
        c = cmul(c, pre);
        dst[0] = {(T)c.re, (T)c.im};
        return;
    }
    cplx<double> Ai[4], Aj[4];
    eval_jones<ORD>(a.bi, fidx, freq, azv, zav, Ai);
    if (a.same_beam) {
        for (int i = 0; i < 4; ++i) Aj[i] = Ai[i];
    } else {
        eval_jones<ORD>(a.bj, fidx, freq, azv, zav, Aj);
    }
    cplx<double> o[4];
    if (!a.pol_sky) {
        const double I = (double)((const T *)flux)[js * a.nfreq + fidx];
        coh_AhB_flux(Ai, Aj, I, o);
    } else {
        // cpu_simulate.py:142-156: the kernels run on A' = flip(A, axis 0)
        const cplx<T> *Cp = (const cplx<T> *)flux + (js * a.nfreq + fidx) * 4;
        cplx<double> C[4];
        for (int i = 0; i < 4; ++i) C[i] = {(double)Cp[i].re, (double)Cp[i].im};
        const cplx<double> Fi[4] = {Ai[2], Ai[3], Ai[0], Ai[1]};
        const cplx<double> Fj[4] = {Aj[2], Aj[3], Aj[0], Aj[1]};
        coh_AhCB(Fi, C, Fj, o);
    }
    if (a.herm == 1) {  // same beam on both sides: o is Hermitian (o_00, o_11 real, o_10 = conj o_01); pre = 1
        dst[0] = {(T)o[0].re, (T)o[3].re};
        dst[1] = {(T)o[1].re, (T)o[1].im};
        return;
    }
    if (a.herm == 2) {  // real Jones matrices on both sides, unpolarized sky: all four products are real
        dst[0] = {(T)o[0].re, (T)o[3].re};
        dst[1] = {(T)o[1].re, (T)o[2].re};
        return;
    }
    for (int r = 0; r < 4; ++r) {
        const cplx<double> v = cmul(o[r], pre);
        dst[r] = {(T)v.re, (T)v.im};
    }
}

// thread <-> (sorted source p, frequency fgi); fgi fastest so a wave reads flux rows contiguously
// and writes its tpol strengths back to back:  cs[p][fgi * tpol + r].
template <typename T, int ORD>
__global__ void k_strengths(StrengthArgs a, const int *__restrict__ Mp, const int *__restrict__ perm,
                            const int *__restrict__ src_idx, const T *__restrict__ az,
                            const T *__restrict__ za, const void *__restrict__ flux,
                            const double *__restrict__ freqs, const int *__restrict__ i0s,
                            const T *__restrict__ fs, cplx<T> *__restrict__ cs, const T *__restrict__ zsrc) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= min((int64_t)*Mp, a.M) * a.nfg) return;
    const int64_t p = idx / a.nfg;
    const int fgi = (int)(idx % a.nfg);
    const int fidx = a.f_first + fgi;
    // type-3 pre-phase exp(i nu btc . x'), x' rebuilt from the sorted grid coordinates
    double dot = 0.0;
    for (int d = 0; d < a.dim; ++d) {
        const double pos = (double)i0s[(int64_t)d * a.M + p] - (double)fs[(int64_t)d * a.M + p];
        dot += a.btc[d] * (pos - 0.5 * a.na[d]) * a.h[d];
    }
    cplx<double> pre = {1.0, 0.0};
    if (dot != 0.0) sincos(freqs[fidx] * dot, &pre.im, &pre.re);
    const int tp = a.herm ? 2 : a.polarized ? 4 : 1;
    cplx<T> *dst = cs + (p * a.nfg + fgi) * tp;
    strength_eval<T, ORD>(a, perm[p], fidx, pre, src_idx, az, za, flux, freqs, dst);
    if (a.wt_k > 0) {  // uniform
        // T_k(t), t = (z - zc) / zh in [-1, 1]: the Chebyshev polynomial of term k (three-term recurrence)
        const double t = ((double)zsrc[perm[p]] - a.wt_zc) * a.wt_inv;
        double sc = t, prev = 1.0;
        for (int i = 1; i < a.wt_k; ++i) {
            const double nx = 2.0 * t * sc - prev;
            prev = sc;
            sc = nx;
        }
        for (int r = 0; r < tp; ++r) dst[r] = {(T)((double)dst[r].re * sc), (T)((double)dst[r].im * sc)};
    }
}

// ---------------------------------------------------------------------------------------------
// Type-1 path (lattice arrays): cpu_nufft2d_type1 (cpu/nufft.py:120-175), set-up
// cpu_simulate.py:661-681, per-slice :964-965,990-992,259-269.
// Visibility of the integer baseline (bx, by) at frequency nu is mode (bx, by) of a type-1
// transform of the sources at angles (tx, ty) = nu * 2 pi B^T topo.  Positions depend on nu, so
// every (source, frequency) pair becomes an entry of its own on that frequency's periodic
// n2 x n2 grid; entries whose footprint crosses the edge get periodic images.  Spread (same
// gather scheme as k_spread2d), pruned FFT keeping the n_modes central outputs, then modes are
// picked and deconvolved -- no gather at non-uniform targets at all.
// ---------------------------------------------------------------------------------------------
constexpr int T1_PAD = 16;  // origins run from -PAD .. n2: bins are offset by PAD cells

struct T1Args {
    int n2, nb1, w, nfg, f_first;
    int64_t cap;       // stride of xyz
    int64_t ecap;      // entry capacity
    int rec;           // bytes per entry record (multiple of 64)
};
// Entry record (rec bytes, 64-B aligned): {int i0x, i0y, ent, 0} then T wx[w], wy[w], zero padding.
// One record per (source, frequency[, periodic image]) in bin order: the scatter writes whole
// 64-B sectors (the earlier split arrays -- 72-B weight rows at arbitrary offsets -- cost 4x their
// size in WRITE_SIZE) and the spread stages a chunk of entries from one contiguous run.
constexpr int T1_HDR = 16;
inline int t1_record_bytes(int w, size_t real_bytes) {
    return (int)((T1_HDR + 2 * (size_t)w * real_bytes + 63) / 64 * 64);
}

// Footprint origin (and first kernel argument) of compacted source p at frequency f.
template <typename T>
__device__ inline void t1_origin(const T1Args &a, const T *__restrict__ xyz, int64_t p, double freq,
                                 int &i0x, int &i0y, double &fx, double &fy) {
    const double inv2pi = 0.5 / M_PI;
    double ux = (double)xyz[p] * freq * inv2pi, uy = (double)xyz[a.cap + p] * freq * inv2pi;
    ux -= floor(ux + 0.5);  // [-0.5, 0.5)
    uy -= floor(uy + 0.5);
    const double px = (ux + 0.5) * a.n2, py = (uy + 0.5) * a.n2;
    i0x = (int)ceil(px - 0.5 * a.w);
    i0y = (int)ceil(py - 0.5 * a.w);
    fx = (double)i0x - px;
    fy = (double)i0y - py;
}

// COUNT = true: histogram of entries per (frequency, bin); false: scatter + tabulate weights.
// W: the kernel width at compile time (9: the default tolerance in fp64; 0: any) -- with it the record is built
// in registers and leaves as 16-byte stores (12 for w = 9) instead of 2 w + padding scattered 8-byte ones.
template <typename T, bool COUNT, int W = 0>
__global__ void k_t1_bin(T1Args a, const int *__restrict__ Mp, const T *__restrict__ xyz,
                         const double *__restrict__ freqs, int *__restrict__ counts,
                         const int *__restrict__ bin_start, int *__restrict__ cursor,
                         unsigned char *__restrict__ recs, T beta, T c4, int *__restrict__ overflow) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= min((int64_t)*Mp, a.cap) * a.nfg) return;
    const int64_t p = idx / a.nfg;
    const int f = (int)(idx % a.nfg);
    int i0x, i0y;
    double fx, fy;
    t1_origin<T>(a, xyz, p, freqs[a.f_first + f], i0x, i0y, fx, fy);
    // periodic images: every origin congruent mod n2 whose footprint reaches [0, n2)
    const int ox[2] = {0, i0x < 0 ? a.n2 : (i0x + a.w > a.n2 ? -a.n2 : 0)};
    const int oy[2] = {0, i0y < 0 ? a.n2 : (i0y + a.w > a.n2 ? -a.n2 : 0)};
    for (int iy = 0; iy < (oy[1] ? 2 : 1); ++iy)
        for (int ix = 0; ix < (ox[1] ? 2 : 1); ++ix) {
            const int jx = i0x + ox[ix], jy = i0y + oy[iy];
            const int bin = (f * a.nb1 + ((jy + T1_PAD) >> BINLOG)) * a.nb1 + ((jx + T1_PAD) >> BINLOG);
            if (COUNT) {
                atomicAdd(&counts[bin], 1);
            } else {
                const int64_t pos = bin_start[bin] + atomicAdd(&cursor[bin], 1);
                if (pos >= a.ecap) {
                    atomicAdd(overflow, 1);
                    continue;
                }
                unsigned char *rec = recs + pos * a.rec;
                *reinterpret_cast<int4 *>(rec) = make_int4(jx, jy, (int)idx, 0);  // ent = p * nfg + f
                T *wr = reinterpret_cast<T *>(rec + T1_HDR);
                if constexpr (W > 0) {
                    constexpr int PER = 16 / (int)sizeof(T);                       // reals per 16-byte store
                    constexpr int NV = (T1_HDR + 2 * W * (int)sizeof(T) + 63) / 64 * 64 / (int)sizeof(T) - T1_HDR / (int)sizeof(T);
                    T v[NV];
#pragma unroll
                    for (int k = 0; k < NV; ++k)
                        v[k] = k < W ? es_eval<T>((T)(fx + k), beta, c4) : k < 2 * W ? es_eval<T>((T)(fy + (k - W)), beta, c4) : T(0);
                    struct alignas(16) V16 { T x[PER]; };
#pragma unroll
                    for (int k = 0; k < NV; k += PER) {
                        V16 t;
#pragma unroll
                        for (int i = 0; i < PER; ++i) t.x[i] = v[k + i];
                        *reinterpret_cast<V16 *>(wr + k) = t;
                    }
                } else {
                    const int nreal = (a.rec - T1_HDR) / (int)sizeof(T);
                    for (int k = 0; k < a.w; ++k) wr[k] = es_eval<T>((T)(fx + k), beta, c4);
                    for (int k = 0; k < a.w; ++k) wr[a.w + k] = es_eval<T>((T)(fy + k), beta, c4);
                    for (int k = 2 * a.w; k < nreal; ++k) wr[k] = T(0);  // whole sectors, no partial writes
                }
            }
        }
}

template <typename T, int ORD>
__global__ void k_t1_strengths(StrengthArgs a, const int *__restrict__ nent, int64_t ecap,
                               const unsigned char *__restrict__ recs, int rec, const int *__restrict__ src_idx,
                               const T *__restrict__ az, const T *__restrict__ za,
                               const void *__restrict__ flux, const double *__restrict__ freqs,
                               cplx<T> *__restrict__ cs) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= min((int64_t)*nent, ecap)) return;  // entries beyond the capacity were dropped (and flagged) by k_t1_bin
    const int id = reinterpret_cast<const int *>(recs + e * rec)[2];
    const int tp = a.herm ? 2 : a.polarized ? 4 : 1;
    strength_eval<T, ORD>(a, id / a.nfg, a.f_first + id % a.nfg, cplx<double>{1.0, 0.0}, src_idx, az, za,
                     flux, freqs, cs + e * tp);
}

// One wave per 16 x 16 cell tile of one frequency plane (lane = (x, y): cells (x | x + 8, y | y + 8)); TP =
// transforms per plane (1, 2 or 4).  The wave walks the entries of the 3 x 3 bins whose footprints reach the tile in
// chunks of 16 -- 0.42 entry visits per cell instead of the 0.75 of an 8 x 8 block per wave:
//   * the origin of an entry is wave-uniform: lane j of the chunk loads entry j's header and the loop over the
//     entries pulls it into scalar registers with v_readlane; the TP strengths are read back from LDS by every
//     lane (a broadcast: LDS returns bytes per lane whether or not the address is shared), which for four cells
//     per lane is a quarter of the LDS bytes per cell of the one-cell version (that one was LDS-bound at 0.14 of
//     HBM peak; all-readlane strengths made an 8 x 16 version VALU-bound instead: 10 readlanes per entry).
//   * only the 2 w kernel weights go through LDS, rows of w + 2 values with a zero at either end: a lane clamps its
//     offsets into [-1, w] (one v_med3 each) instead of testing them, and reads x, y and y + 8.
// Every cell of the plane is written exactly once; no atomics.
template <typename T, int TP>
__global__ __launch_bounds__(SPREAD_THREADS) void k_t1_spread(
    T1Args a, const unsigned char *__restrict__ recs, const int *__restrict__ bin_start,
    const cplx<T> *__restrict__ cs, cplx<T> *__restrict__ grid) {
    constexpr int KW = MAX_W + 2;
    constexpr int TL = BINLOG + 1;  // tile = 16 x 16 cells
    __shared__ T s_kw[SPREAD_THREADS / 64][SPREAD_CHUNK][2][KW];
    __shared__ cplx<T> s_str[SPREAD_THREADS / 64][SPREAD_CHUNK][TP];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int bx2 = blockIdx.x * 4 + wave, by2 = blockIdx.y, f = blockIdx.z;
    if (bx2 >= (a.n2 >> TL)) return;  // wave-uniform; the waves of a workgroup share nothing
    const int tx0 = bx2 << TL, ty0 = by2 << TL;  // the tile's first cell (uniform)
    const int cx = tx0 + (lane & 7), cy = ty0 + (lane >> 3);
    T ar[4][TP], ai[4][TP];  // cells (cx, cy), (cx + 8, cy), (cx, cy + 8), (cx + 8, cy + 8)
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int q = 0; q < TP; ++q) ar[h][q] = ai[h][q] = T(0);
    const int w = a.w;
    // zero guards of the weight rows (slots 0 and w + 1), written once
    for (int e = lane; e < SPREAD_CHUNK * 4; e += 64) s_kw[wave][e >> 2][(e >> 1) & 1][(e & 1) ? w + 1 : 0] = T(0);
    const int bxl = ((bx2 << TL) - w + 1 + T1_PAD) >> BINLOG, bxh = ((bx2 << TL) + 15 + T1_PAD) >> BINLOG;
    const int byl = ((by2 << TL) - w + 1 + T1_PAD) >> BINLOG, byh = ((by2 << TL) + 15 + T1_PAD) >> BINLOG;
    // The chunks of the <= 3 bin rows form one sequence; chunk c + 1 is requested (global -> registers) before chunk
    // c is accumulated, so its loads fly under the accumulation.
    constexpr int NWV = (2 * MAX_W + 3) / 4;  // weights per staging lane (4 lanes per entry)
    int yb = byl, s1 = 0, base = 0;
    auto open_row = [&]() {  // [base, s1) of bin row yb; (entries beyond the capacity were dropped and flagged by k_t1_bin)
        const int rowb = (f * a.nb1 + yb) * a.nb1;
        base = __builtin_amdgcn_readfirstlane(bin_start[rowb + bxl]);
        s1 = __builtin_amdgcn_readfirstlane((int)min((int64_t)bin_start[rowb + bxh + 1], a.ecap));
    };
    auto next_chunk = [&](int &cb, int &cn) {  // false: no chunk left
        while (base >= s1) {
            if (yb > byh) return false;
            open_row();
            ++yb;
        }
        cb = base;
        cn = min(SPREAD_CHUNK, s1 - base);
        base += cn;
        return true;
    };
    int2 hdr_n = make_int2(0, 0);
    cplx<T> sv_n = {T(0), T(0)};
    T wv_n[NWV];
    auto request = [&](int cb, int cn) {
        const unsigned char *rb = recs + (int64_t)cb * a.rec;
        if (lane < cn) hdr_n = *reinterpret_cast<const int2 *>(rb + (int64_t)lane * a.rec);
        if (lane < cn * TP) sv_n = cs[(int64_t)cb * TP + lane];
        const int j = lane >> 2;  // weights: 4 lanes per entry, every fourth value each
        const T *wr = reinterpret_cast<const T *>(rb + (int64_t)j * a.rec + T1_HDR);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int k = (lane & 3) + 4 * i;
            wv_n[i] = j < cn && k < 2 * w ? wr[k] : T(0);
        }
    };
    int cb = 0, cn = 0;
    bool have = next_chunk(cb, cn);
    if (have) request(cb, cn);
    while (have) {
        const int n = cn;
        const int2 hdr = hdr_n;  // staging register: lane j = entry j's origin
        if (lane < n * TP) s_str[wave][lane / TP][lane % TP] = sv_n;
        {
            const int j = lane >> 2;
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const int k = (lane & 3) + 4 * i;
                if (j < n && k < 2 * w) s_kw[wave][j][k >= w][(k >= w ? k - w : k) + 1] = wv_n[i];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        have = next_chunk(cb, cn);
        if (have) request(cb, cn);
        for (int j = 0; j < n; ++j) {
            const int ox = __builtin_amdgcn_readlane(hdr.x, j), oy = __builtin_amdgcn_readlane(hdr.y, j);
            // (Skipping, with scalar branches on the wave-uniform origin, the quadrants a footprint misses -- 1.8 of 4 are met
            // on average -- was measured slower: 1.63 against 1.22 ms per C3 time step; the branches break the schedule.)
            // offsets clamped into [-1, w]: the guards at either end of a row are zero
            const int dx = cx - ox, dy = cy - oy;
            auto clamp = [&](int v) {  // median(v, -1, w): one instruction (the compiler cannot prove -1 <= w for min(max()))
                int r;
                asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(v), "s"(w));
                return r + 1;
            };
            const T wx0 = s_kw[wave][j][0][clamp(dx)], wx1 = s_kw[wave][j][0][clamp(dx + 8)];
            const T wy0 = s_kw[wave][j][1][clamp(dy)], wy1 = s_kw[wave][j][1][clamp(dy + 8)];
            const T wt[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
            for (int q = 0; q < TP; ++q) {
                const cplx<T> cv = s_str[wave][j][q];  // broadcast read
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    ar[h][q] += cv.re * wt[h];
                    ai[h][q] += cv.im * wt[h];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the rows are rewritten by the next chunk
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const int64_t plane = (int64_t)a.n2 * a.n2;
    cplx<T> *o = grid + (int64_t)f * TP * plane + (int64_t)cy * a.n2 + cx;
#pragma unroll
    for (int q = 0; q < TP; ++q)
#pragma unroll
        for (int h = 0; h < 4; ++h) o[q * plane + (int64_t)(h >> 1) * 8 * a.n2 + (h & 1) * 8] = {ar[h][q], ai[h][q]};
}

// The same tile walk with the accumulation on the matrix pipe (fp64).  A footprint's weights are separable, so the update
// of one 16 x 16 tile by the entries e that reach it is a matrix product per real component c of the strengths,
//     G_c[y][x] += sum_e (s_{e,c} wy_e[y]) wx_e[x]      --   A[y][e] = s_{e,c} wy_e[y] (16 x 4),  B[e][x] = wx_e[x] (4 x 16),
// v_mfma_f64_16x16x4_f64: four entries per instruction, 2 TP instructions (the real and imaginary parts of the TP
// transforms) per four entries.  fp64 MFMA has the vector pipe's flop rate on gfx950, and seven eighths of the products
// are by the zero guards here too; what it removes is everything around the multiply-adds -- the vector version spends 38
// instructions per (entry, tile) visit, 16 of them multiply-adds; here a visit is one MFMA (64 cycles) plus a quarter of
// two LDS reads, two clamps and 2 TP multiplies, issued beside it.  C3 lattice path: 1.22 -> see profiles/MEASUREMENTS.md.
// Operand layout (as k_spread2d_mm): lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15] and D[(l >> 4) + 4 r][l & 15], r < 4.
// (The waves-per-SIMD bound is what makes the compiler keep the accumulators in vector registers: without it they
// travel to accumulator registers and back around every group of MFMAs -- 64 moves per four instructions, and moves
// issue on the same pipe: 25 vector instructions per MFMA, the kernel no faster than the vector version.)
template <int TP>
__global__ __launch_bounds__(SPREAD_THREADS, 4) void k_t1_spread_mm(
    T1Args a, const unsigned char *__restrict__ recs, const int *__restrict__ bin_start,
    const cplx<double> *__restrict__ cs, cplx<double> *__restrict__ grid) {
    using T = double;
    using d4 = double __attribute__((ext_vector_type(4)));
    constexpr int KW = MAX_W + 2;
    constexpr int TL = BINLOG + 1;  // tile = 16 x 16 cells
    __shared__ T s_kw[SPREAD_THREADS / 64][SPREAD_CHUNK][2][KW];
    __shared__ cplx<T> s_str[SPREAD_THREADS / 64][SPREAD_CHUNK][TP];
    __shared__ int s_org[SPREAD_THREADS / 64][SPREAD_CHUNK][2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int bx2 = blockIdx.x * 4 + wave, by2 = blockIdx.y, f = blockIdx.z;
    if (bx2 >= (a.n2 >> TL)) return;  // wave-uniform; the waves of a workgroup share nothing
    const int tx0 = bx2 << TL, ty0 = by2 << TL;
    const int li = lane & 15, g = lane >> 4;  // operand row / column, and the entry of a group of four
    d4 acc[2 * TP];
#pragma unroll
    for (int c = 0; c < 2 * TP; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
    const int w = a.w;
    // zero guards of the weight rows (slots 0 and w + 1) and zero strengths / origins everywhere: the slots a short chunk
    // leaves untouched enter the products with zero strengths and finite weights
    for (int e = lane; e < SPREAD_CHUNK * 2 * KW; e += 64) (&s_kw[wave][0][0][0])[e] = T(0);
    for (int e = lane; e < SPREAD_CHUNK * TP; e += 64) (&s_str[wave][0][0])[e] = {T(0), T(0)};
    if (lane < SPREAD_CHUNK * 2) (&s_org[wave][0][0])[lane] = 0;
    const int bxl = (tx0 - w + 1 + T1_PAD) >> BINLOG, bxh = (tx0 + 15 + T1_PAD) >> BINLOG;
    const int byl = (ty0 - w + 1 + T1_PAD) >> BINLOG, byh = (ty0 + 15 + T1_PAD) >> BINLOG;
    constexpr int NWV = (2 * MAX_W + 3) / 4;  // weights per staging lane (4 lanes per entry)
    int yb = byl, s1 = 0, base = 0;
    auto open_row = [&]() {
        const int rowb = (f * a.nb1 + yb) * a.nb1;
        base = __builtin_amdgcn_readfirstlane(bin_start[rowb + bxl]);
        s1 = __builtin_amdgcn_readfirstlane((int)min((int64_t)bin_start[rowb + bxh + 1], a.ecap));
    };
    auto next_chunk = [&](int &cb, int &cn) {
        while (base >= s1) {
            if (yb > byh) return false;
            open_row();
            ++yb;
        }
        cb = base;
        cn = min(SPREAD_CHUNK, s1 - base);
        base += cn;
        return true;
    };
    int2 hdr_n = make_int2(0, 0);
    cplx<T> sv_n = {T(0), T(0)};
    T wv_n[NWV];
    // (every load is unconditional -- indices clamped into the chunk, the value selected afterwards: as conditional loads
    // each sat in an exec-masked region of its own with a branch around it, and vector instructions of the staging
    // serialise with the MFMAs on the SIMD)
    const int nwv = (2 * w + 3) >> 2;  // uniform
    auto request = [&](int cb, int cn) {  // cn >= 1
        const unsigned char *rb = recs + (int64_t)cb * a.rec;
        hdr_n = *reinterpret_cast<const int2 *>(rb + (int64_t)min(lane, cn - 1) * a.rec);
        sv_n = cs[(int64_t)cb * TP + min(lane, cn * TP - 1)];
        const int j = lane >> 2;
        const T *wr = reinterpret_cast<const T *>(rb + (int64_t)min(j, cn - 1) * a.rec + T1_HDR);
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int k = (lane & 3) + 4 * i;
            if (i < nwv) wv_n[i] = wr[min(k, 2 * w - 1)];  // uniform; used (or not) when the chunk is staged, not here
        }
    };
    int cb = 0, cn = 0;
    bool have = next_chunk(cb, cn);
    if (have) request(cb, cn);
    while (have) {
        const int n = cn;
        if (lane < SPREAD_CHUNK * TP) s_str[wave][lane / TP][lane % TP] = lane < n * TP ? sv_n : cplx<T>{T(0), T(0)};  // zero beyond the chunk's count
        if (lane < n) {
            s_org[wave][lane][0] = hdr_n.x - tx0;  // origin relative to the tile
            s_org[wave][lane][1] = hdr_n.y - ty0;
        }
        {
            const int j = lane >> 2;
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const int k = (lane & 3) + 4 * i;
                if (j < n && k < 2 * w) s_kw[wave][j][k >= w][(k >= w ? k - w : k) + 1] = wv_n[i];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        have = next_chunk(cb, cn);
        if (have) request(cb, cn);
        auto clamp = [&](int v) {  // median(v, -1, w) + 1: the guards at either end of a row are zero
            int r;
            asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(v), "s"(w));
            return r + 1;
        };
        const int nq = (n + 3) >> 2;
        for (int q = 0; q < nq; ++q) {
            const int e = 4 * q + g;  // this lane's entry (slots beyond n: zero strengths)
            const int ox = s_org[wave][e][0], oy = s_org[wave][e][1];
            const T wy = s_kw[wave][e][1][clamp(li - oy)];  // A: row y = li
            const T wx = s_kw[wave][e][0][clamp(li - ox)];  // B: column x = li
#pragma unroll
            for (int t = 0; t < TP; ++t) {
                const cplx<T> sv = s_str[wave][e][t];
                acc[2 * t] = __builtin_amdgcn_mfma_f64_16x16x4f64(sv.re * wy, wx, acc[2 * t], 0, 0, 0);
                acc[2 * t + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sv.im * wy, wx, acc[2 * t + 1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the rows are rewritten by the next chunk
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // results: lane l, register r = cell (x = tx0 + (l & 15), y = ty0 + (l >> 4) + 4 r): 16 lanes write 256 contiguous bytes
    const int64_t plane = (int64_t)a.n2 * a.n2;
    cplx<T> *o = grid + (int64_t)f * TP * plane + (int64_t)(ty0 + g) * a.n2 + tx0 + li;
#pragma unroll
    for (int t = 0; t < TP; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t * plane + (int64_t)(4 * r) * a.n2] = {acc[2 * t][r], acc[2 * t + 1][r]};
}

// vis[f][pol][k] = X_{f,pol}[bx_k][by_k] / (psi_hat(bx) psi_hat(by)); flipped baselines take the
// negated mode and are conjugated (cpu_simulate.py:259,298).  X is stored [plane][lx][ly].
template <typename T>
__global__ void k_t1_pick(const cplx<T> *__restrict__ X, int no, int P, int cnt, int nfg, int tp,
                          const int *__restrict__ blx, const int *__restrict__ bly, int64_t N,
                          const int *__restrict__ bl_idx, const signed char *__restrict__ flip,
                          const T *__restrict__ dec, cplx<T> *__restrict__ out,
                          int64_t out_fg_stride, int64_t p0, int64_t p1, int64_t p2, int64_t p3, bool accumulate,
                          bool herm, bool tflip) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * nfg) return;
    const int f = (int)(idx / N);
    const int64_t kl = idx % N;
    const int64_t k = bl_idx ? bl_idx[kl] : kl;
    const bool fl = flip && flip[kl];
    const int mx = fl ? -blx[k] : blx[k], my = fl ? -bly[k] : bly[k];
    const int lx = mx + no / 2, ly = my + no / 2;
    const T d = dec[lx] * dec[ly];
    // tflip (reference_compat = 0): a flipped baseline's block goes to the feed-transposed slots (V_ij(-b)^H)
    const bool sw = fl && tflip && tp != 1;
    const int64_t pol[4] = {p0, sw ? p2 : p1, sw ? p1 : p2, p3};
    if (herm) {
        // Hermitian packing (see k_interp): planes T1 = F[c_00 + i c_11], T2 = F[c_01]; the mirror mode
        // (-mx, -my) is on the grid too and shares the deconvolution factor (psi_hat is even)
        const int lxm = -mx + no / 2, lym = -my + no / 2;
        const int64_t rows = (int64_t)P * cnt;
        const cplx<T> *X1 = X + ((int64_t)f * 2) * rows * no, *X2 = X1 + rows * no;
        const cplx<T> Pp = X1[(int64_t)out_pos(lx, P, cnt) * no + ly], Mm = X1[(int64_t)out_pos(lxm, P, cnt) * no + lym];
        const cplx<T> C = X2[(int64_t)out_pos(lx, P, cnt) * no + ly], D = X2[(int64_t)out_pos(lxm, P, cnt) * no + lym];
        const T h = T(0.5) * d;
        cplx<T> v[4];
        v[0] = {h * (Pp.re + Mm.re), h * (Pp.im - Mm.im)};   // (P + conj M) / 2
        v[3] = {h * (Pp.im + Mm.im), -h * (Pp.re - Mm.re)};  // (P - conj M) / 2i
        v[1] = {d * C.re, d * C.im};
        v[2] = {d * D.re, -d * D.im};                        // conj D
        for (int r = 0; r < 4; ++r) {
            if (fl) v[r].im = -v[r].im;
            cplx<T> &o = out[(int64_t)f * out_fg_stride + pol[r] + k];
            o = accumulate ? cplx<T>{o.re + v[r].re, o.im + v[r].im} : v[r];
        }
        return;
    }
    for (int r = 0; r < tp; ++r) {
        // rows (lx) are stored residue-major (DimGeom::out_pos), the contiguous ly in natural order
        cplx<T> v = X[(((int64_t)f * tp + r) * ((int64_t)P * cnt) + out_pos(lx, P, cnt)) * no + ly];
        v = {v.re * d, fl ? -v.im * d : v.im * d};
        cplx<T> &o = out[(int64_t)f * out_fg_stride + pol[r] + k];
        o = accumulate ? cplx<T>{o.re + v.re, o.im + v.im} : v;
    }
}

// Brute-force type-3 sum on the device (independent checker; fp64 accumulation).
template <typename T>
__global__ void k_nudft_direct(int dim, int64_t M, const T *__restrict__ x, const T *__restrict__ y,
                               const T *__restrict__ z, const cplx<T> *__restrict__ c, int ntrans,
                               int64_t N, const T *__restrict__ s, const T *__restrict__ t,
                               const T *__restrict__ u, cplx<T> *__restrict__ out) {
    const int64_t k = blockIdx.x;
    const int tr = blockIdx.y;
    if (k >= N) return;
    const double sk = s[k], tk = dim > 1 ? (double)t[k] : 0.0, uk = dim > 2 ? (double)u[k] : 0.0;
    double ar = 0.0, ai = 0.0;
    for (int64_t j = threadIdx.x; j < M; j += blockDim.x) {
        double ph = sk * (double)x[j];
        if (dim > 1) ph += tk * (double)y[j];
        if (dim > 2) ph += uk * (double)z[j];
        double sn, cs;
        sincos(ph, &sn, &cs);
        const cplx<T> cv = c[(int64_t)tr * M + j];
        ar += (double)cv.re * cs - (double)cv.im * sn;
        ai += (double)cv.re * sn + (double)cv.im * cs;
    }
    __shared__ double rr[256], ri[256];
    rr[threadIdx.x] = ar;
    ri[threadIdx.x] = ai;
    __syncthreads();
    for (int off = blockDim.x / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            rr[threadIdx.x] += rr[threadIdx.x + off];
            ri[threadIdx.x] += ri[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(int64_t)tr * N + k] = {(T)rr[0], (T)ri[0]};
}

// ---------------------------------------------------------------------------------------------
// Engine
// ---------------------------------------------------------------------------------------------
struct SimBase {
    virtual ~SimBase() = default;
    virtual void set_sources(int64_t nsrc, int nfreq, const void *eq, const void *flux, int pol_sky,
                             int on_device) = 0;
    virtual void set_times(int ntimes, const double *rot) = 0;
    virtual void set_topo(int ntimes, int64_t nsrc, const void *topo, int on_device) = 0;
    virtual void set_astrom(int ntimes, const double *astrom) = 0;
    virtual void set_freqs(int nfreq, const double *freqs) = 0;
    virtual void set_array(const double *R, int64_t nbls, const double *bls, int coplanar) = 0;
    virtual void set_array_type1(const double *basis, int64_t nbls, const int *bls_int, int n_modes) = 0;
    virtual void set_nbeams(int n) = 0;
    virtual void set_beam_airy(int b, double diameter, const double *jones_scale, double power_scale) = 0;
    virtual void set_reference_compat(int on) = 0;
    virtual void set_beam_table(int b, int nfreq_tab, int nza, int naz, double za_max,
                                const void *table, int order) = 0;
    virtual void set_beam_pairs(int npairs, const int *bi, const int *bj, const int64_t *off,
                                const int *idx, const signed char *flipped) = 0;
    virtual void set_basis(int nant, int K, int nfreq, const void *coefs, const int *ant1,
                           const int *ant2) = 0;
    virtual void set_chunking(int nchunks, double source_buffer) = 0;
    virtual void run(int t0, int t1, int f0, int f1, void *out, int out_on_device) = 0;
    // Host destination of the next run (fv_sim_run_into): `out` is then a block INSIDE a larger array -- channel f of
    // the block starts f * out_f_stride elements after `out` (0: the block is contiguous) -- and with out_shared other
    // processes write the rest of that array (a sharded run's ranks filling one shared result): the pinning helper must
    // then neither write to it nor register more than the block's own runs.  Reset by every run.
    int64_t out_f_stride = 0;
    int out_shared = 0;
    virtual void sync() = 0;
    virtual void stats(double *v, int n) = 0;
    virtual void reset_stats() = 0;
    virtual void enable_timing(int on) = 0;
    virtual void timing(double *ms, int n) = 0;
};

enum { TM_SPREAD = 0, TM_FFT, TM_INTERP, TM_STRENGTHS, TM_PREP, TM_COUNT };

template <typename T>
class Sim : public SimBase {
    int device;
    hipStream_t stream = nullptr;
    double eps, sigma;       // sigma: the caller's upsampling factor, 0 = chosen per run (see run())
    double sigma_run = 2.0;  // the factor the last run used
    bool polarized;
    int tpol;

    int64_t nsrc = 0;
    int nfreq_cat = 0;
    bool pol_sky = false;
    DevBuf d_eq, d_flux;

    std::vector<Rot9> rots;
    int ntimes_topo = 0;  // > 0: per-time topocentric unit vectors were supplied instead
    std::vector<Astrom> astroms;  // non-empty: per-time astrometry contexts, applied on the device (k_astrom_topo)
    DevBuf d_topo;        // (ntimes, 3, nsrc) T
    std::vector<double> freqs;
    DevBuf d_freqs;

    Rot9 rplane{};
    int64_t nbls = 0;
    bool coplanar = true;
    std::vector<double> h_bls;  // (3, nbls) seconds
    bool order_pairs = true;    // set_beam_pairs visits a pair's baselines in (u, v) order
    DevBuf d_bls;               // (3, nbls) T

    int beam_order = 1;  // interpolation order of the tabulated beams (1 or 3)
    struct Beam {
        int kind = -1;
        double diameter = 0;
        double js[8] = {1, 0, 1, 0, 1, 0, 1, 0}, ps = 1;  // Airy: Jones-slot factors, power factor
        int nfreq_tab = 0, nza = 0, naz = 0;
        double za_max = 0;
        std::unique_ptr<DevBuf> table;
        bool real_valued = true;  // every Jones entry has zero imaginary part (Airy; tables are scanned at upload)
    };
    std::vector<Beam> beams;

    struct Pair {
        int bi, bj;
        int64_t n;
        bool trivial;  // all baselines in order, nothing flipped
        std::unique_ptr<DevBuf> idx, flip;
        // the list in the CALLER's order (usually increasing baseline index): what the lattice path's mode pick walks --
        // its reads are a few thousand distinct modes of a plane that sits in L2 whatever the order, its WRITES are 16 bytes
        // per (baseline, product) and want neighbouring threads on neighbouring baselines
        bool trivial0 = false;
        std::unique_ptr<DevBuf> idx0, flip0;
        // Redundant baselines: runs of the (u, v)-ordered list whose sign-adjusted vectors agree (build_unique) are ONE
        // target of the gather.  h_idx / h_flip: the list as visited (host copy); ustart: nu + 1 run starts (device).
        std::vector<int> h_idx, h_ustart;
        std::vector<signed char> h_flip;
        bool sorted = false;  // the list is visited in (u, v) order
        std::unique_ptr<DevBuf> ustart, upairs;  // run starts; (packed runs) pairs of runs b / -b that share one gather item
        int64_t nu = 0, nitems = 0;               // runs; gather items (= runs unless paired)
        double utol = -1.0;
        int udims = 3;  // components compared when runs were built (2 under height terms: b_z is per member there)
        int upairs_herm = -1;
        double btc[3], B[3];   // tight box of its (sign-adjusted) baselines: centre, half-width [s]
        double Bs[3];          // half-width of the box made symmetric about 0
        int herm = 0;          // this run packs its strengths into two transforms: 1 Hermitian, 2 all real (per run)
        bool mirror = false;   // this run also gathers at the mirror targets -b (exact eigenbeam (l, k) terms)
        const double *box_c() const { return herm || mirror ? zero3 : btc; }
        const double *box_B() const { return herm || mirror ? Bs : B; }
    };
    static constexpr double zero3[3] = {0.0, 0.0, 0.0};
    std::vector<Pair> pairs;
    // Source-axis chunking (reference cpu_simulate.py:939: `for chunk in range(nchunks)` inside the time
    // loop, visibilities accumulate with +=): per-time scratch is sized by one chunk, the catalog stays
    // resident.  source_buffer = fraction of a chunk the above-horizon arrays can hold (matvis sizes its
    // buffers the same way and raises when a chunk has more sources above the horizon).
    int src_chunks = 1;
    double source_buffer = 1.0;
    int nbasis = 0;  // > 0: eigenbeam mode
    // type-1 (lattice) mode
    bool type1 = false;
    int t1_nmodes = 0;
    DevBuf d_blint;  // (2, nbls) int
    DevBuf t1_meta[2], t1_binstart[2], t1_rec[2], t1_cs, t1_dec;  // [2]: pipelined (time, batch) units
    std::unique_ptr<Nufft3<T>> t1fft;
    DevBuf d_coefs, d_ant1, d_ant2;

    // Per-time scratch lives in a lane.  Small problems run consecutive time steps on two lanes
    // (two streams) so that one step's launch ramps and tails overlap the other's kernels.
    struct Lane {
        hipStream_t stream = nullptr;
        bool own_stream = false;
        hipEvent_t done = nullptr;
        hipEvent_t prep_done = nullptr, heavy_done = nullptr;  // pipelined mode (see run())
        bool heavy_pending = false;
        std::unique_ptr<Nufft3<T>> nufft;
        std::unique_ptr<Nufft3<T>> nufft_l[2];  // the plans of the LIGHT height terms (see Sim::wt_k0): looser tolerances, sigma = 1.25
        DevBuf d_xyz, d_az, d_za, d_srcidx, d_blockcnt, d_blockoff, d_scan_tot, d_scan_off, d_enu;
        int binned_ti = -1;
        int64_t binned_serial = -1;
        int binned_ti_l[2] = {-1, -1};  // the same for nufft_l
        int64_t binned_serial_l[2] = {-1, -1};
    };
    Lane lanes[4];  // [2], [3]: second pair of the gang mode (see run())
    int lane_mode = -1;       // 0 one stream per lane, 1 pipelined, 2 pipelined gangs: what the lanes last ran as
    int64_t lane_serial = 0;  // units processed in that mode (lane rotation continues across runs)
    hipStream_t prep_stream = nullptr;  // low priority: per-time preparation of the next step
    // Host output of a large run (drain_*): the caller's array is pinned in place while the GPU computes, and every
    // finished time step leaves through this stream beside the later steps' kernels.
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> drain_events;  // "time step(s) finished", reused from run to run
    struct DrainItem {
        hipEvent_t ev;
        int t, n;  // time steps [t, t + n) of the run's block
    };
    hipEvent_t ev_start = nullptr;
    DevBuf d_out, d_mhist;
    // sticky device-side error counters, read at every host synchronisation point (check_errors):
    // [0] sources outside the planned box or with NaN coordinates (k_bin_count), [1] type-1 entries
    // dropped because the entry buffers overflowed (k_t1_bin), [2] above-horizon sources that did not
    // fit source_buffer x chunk size (k_horizon_compact), [3] footprint columns missing from a column plan (k_interp)
    DevBuf d_err;
    std::vector<std::pair<int, double>> mhist_log;  // (time index, transforms spread) per processed time

    // stats / timing
    double st[24] = {0};
    int timing_level = 0;  // 1: spread only, sampled (events ride on the dispatches); 3: the same on every spread launch; 2: every kernel family
    int64_t targets_serial = 1;  // version of the device-side target data (baselines, frequencies, pair lists)
    struct Ev {
        hipEvent_t a, b;
        int kind;
    };
    std::vector<Ev> ev_pool;
    size_t ev_used = 0;
    double tm[TM_COUNT] = {0};
    // level 1 attaches events to the spread launches of one time step in TIMING_STRIDE (16; the 9th of
    // each 16, a steady-state one rather than the first after the run's set-up) only (all
    // frequency groups of that step, so the sample is representative): even dispatch-attached
    // events leave ~5-8 us of idle queue on either side of a launch
    static constexpr int TIMING_STRIDE = 16;
    double spread_timed = 0;

    int dim() const { return coplanar ? 2 : 3; }
    // Height terms ("w-term expansion", run()): a non-coplanar array whose heights are small against the wavelength --
    // every surveyed real array: centimetres to decimetres after the plane fit -- does not need a third grid dimension.
    // exp(i z s_z), z the sources' height coordinate in [zc - zh, zc + zh], s_z = nu b_z, is expanded in Chebyshev
    // polynomials of t = (z - zc) / zh (Jacobi - Anger: exp(i a t) = J_0(a) + 2 sum_k i^k J_k(a) T_k(t)):
    //     V(s) = sum_k  exp(i zc s_z) c_k(zh s_z)  F_k(s_x, s_y),   c_0 = J_0, c_k = 2 i^k J_k,   F_k = 2-D transform of c_j T_k(t_j),
    // K terms with 2 (a / 2)^K / K! <= eps / 10, a = zh max|s_z| (|J_k(a)| <= (a / 2)^k / k!): K 2-D transforms (7 for 3 cm of
    // scatter at 200 MHz; the Taylor series about zc this replaced needed a^K / K! <= eps / 10: 8) with all of the 2-D
    // machinery (Hermitian packing, column plan, source disc) instead of a 3-D grid whose third dimension is all kernel
    // width (16 planes for a source range of 1.4 cells) plus a z-pass.  The terms carry the transform's relative error
    // each, |T_k| <= 1, summed with weights |c_k| (sum <= 2 e^{a/2} - 1): the 2-D plans run at eps / (2 e^{a/2} - 1).
    // Taken while K <= 16 (|b_z| up to metres); beyond, or with FFTVIS_HIP_NO_WTERM=1, the 3-D transform runs.
    int wt_K = 0;   // terms of the current run (0: no expansion)
    double wt_zc = 0.0, wt_zh = 0.0, wt_a = 0.0;
    // Light terms: term k enters with weight |c_k| <= 2 (a / 2)^k / k!, so the higher terms need far less than the run's
    // tolerance -- with a = 0.35 (3 cm of scatter at 200 MHz) |c_2| = 0.03, |c_4| = 8e-5.  The terms k >= wt_k0 run on a
    // second plan at wt_eps_l = 0.3 eps / sum_{k >= k0} |c_k| and sigma = 1.25 (a grid of 0.39 x the cells: the FFT passes
    // are the bulk of a term), chosen as the smallest k0 whose tolerance is above that sigma's floor by a decade; fp64,
    // runs at sigma = 2 on large grids only.  wt_k0 = 0: every term on the run's own plan.  FFTVIS_HIP_NO_WTERM_LIGHT=1.
    // Where the tail of the light terms can take a tolerance of 1e-4 or looser (a kernel of 8 cells instead of 12: the
    // gather of a light term is most of its time) the light terms split in two classes, [k0, k1) and [k1, K), with half
    // of the budget each; wt_k1 = wt_K: one class.
    int wt_k0 = 0, wt_k1 = 0;
    double wt_eps_l[2] = {0.0, 0.0};
    int run_D = 2;  // dimensions of the current run's transforms

    size_t ev_slot(int kind) {
        if (ev_used == ev_pool.size()) {
            Ev e;
            FV_HIP(hipEventCreate(&e.a));
            FV_HIP(hipEventCreate(&e.b));
            e.kind = kind;
            ev_pool.push_back(e);
        }
        ev_pool[ev_used].kind = kind;
        return ev_used++;
    }
    size_t ev_begin(int kind, hipStream_t st_) {
        if (timing_level != 2) return (size_t)-1;
        if (ev_used == ev_pool.size()) {
            Ev e;
            FV_HIP(hipEventCreate(&e.a));
            FV_HIP(hipEventCreate(&e.b));
            e.kind = kind;
            ev_pool.push_back(e);
        }
        ev_pool[ev_used].kind = kind;
        FV_HIP(hipEventRecord(ev_pool[ev_used].a, st_));
        return ev_used++;
    }
    void ev_end(size_t i, hipStream_t st_) {
        if (i == (size_t)-1) return;
        FV_HIP(hipEventRecord(ev_pool[i].b, st_));
    }
    void ev_collect() {
        for (size_t i = 0; i < ev_used; ++i) {
            float ms = 0;
            FV_HIP(hipEventElapsedTime(&ms, ev_pool[i].a, ev_pool[i].b));
            tm[ev_pool[i].kind] += ms;
        }
        ev_used = 0;
    }

   public:
    Sim(int device_, double eps_, double sigma_, int polarized_)
        : device(device_), eps(eps_), sigma(sigma_), polarized(polarized_ != 0),
          tpol(polarized_ ? 4 : 1) {
        FV_HIP(hipSetDevice(device));
        int prio_least = 0, prio_greatest = 0;
        FV_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        FV_HIP(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio_greatest));
        lanes[0].stream = stream;
        // the second lane's stream has the main stream's priority: two free-running lanes then share the dispatcher like two
        // processes do (with a lower priority the second lane only ever filled the first one's tails)
        FV_HIP(hipStreamCreateWithPriority(&lanes[1].stream, hipStreamNonBlocking,
                                           std::getenv("FFTVIS_HIP_LANE1_LOW") ? (prio_least + prio_greatest) / 2 : prio_greatest));
        lanes[1].own_stream = true;
        // (streams of a third and fourth free-running lane are made on demand, run(): streams share the few hardware
        // queues, and two more of them at creation put the main stream and the low-priority preparation stream of the
        // pipelined small-grid mode on one queue -- C2 1.43 -> 3.6 ms per step)
        FV_HIP(hipStreamCreateWithPriority(&prep_stream, hipStreamNonBlocking, prio_least));
        for (Lane &L : lanes) FV_HIP(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        for (Lane &L : lanes) {
            FV_HIP(hipEventCreateWithFlags(&L.prep_done, hipEventDisableTiming));
            FV_HIP(hipEventCreateWithFlags(&L.heavy_done, hipEventDisableTiming));
        }
        FV_HIP(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
        d_err.reserve(4 * sizeof(int));
        FV_HIP(hipMemsetAsync(d_err.p, 0, 4 * sizeof(int), stream));
        for (int i = 0; i < 9; ++i) rplane.m[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    ~Sim() override {
        (void)hipSetDevice(device);
        for (Lane &L : lanes) {
            L.nufft.reset();
            L.nufft_l[0].reset();
            L.nufft_l[1].reset();
            if (L.done) (void)hipEventDestroy(L.done);
            if (L.prep_done) (void)hipEventDestroy(L.prep_done);
            if (L.heavy_done) (void)hipEventDestroy(L.heavy_done);
            if (L.own_stream && L.stream) (void)hipStreamDestroy(L.stream);
        }
        if (prep_stream) (void)hipStreamDestroy(prep_stream);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (hipEvent_t e : drain_events) (void)hipEventDestroy(e);
        if (ev_start) (void)hipEventDestroy(ev_start);
        for (auto &e : ev_pool) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        if (stream) (void)hipStreamDestroy(stream);
    }

    void upload(DevBuf &dst, const void *src, size_t bytes, int on_device) {
        dst.reserve(std::max<size_t>(bytes, 16));
        if (bytes)
            FV_HIP(hipMemcpyAsync(dst.p, src, bytes,
                                  on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        FV_HIP(hipStreamSynchronize(stream));
    }

    void set_sources(int64_t n, int nfreq, const void *eq, const void *flux, int ps,
                     int on_device) override {
        FV_HIP(hipSetDevice(device));
        FV_REQUIRE(n >= 0 && nfreq >= 1, "bad catalog shape");
        FV_REQUIRE(n < (int64_t)1 << 31, "catalog too large for 32-bit source indices");
        FV_REQUIRE(!ps || polarized, "polarized sky needs a polarized engine (cpu/utils.py:56-66)");
        nsrc = n;
        nfreq_cat = nfreq;
        pol_sky = ps != 0;
        upload(d_eq, eq, sizeof(T) * 3 * n, on_device);
        upload(d_flux, flux, (pol_sky ? sizeof(T) * 8 : sizeof(T)) * (size_t)n * nfreq, on_device);
    }
    void set_astrom(int ntimes, const double *astrom) override {
        mhist_log.clear();
        ntimes_topo = 0;
        rots.assign(ntimes, Rot9{{1, 0, 0, 0, 1, 0, 0, 0, 1}});  // the horizon kernels then read finished ENU vectors
        astroms.resize(ntimes);
        std::memcpy(astroms.data(), astrom, sizeof(Astrom) * (size_t)ntimes);
        for (const Astrom &a : astroms) {
            FV_REQUIRE(a.em > 0 && a.bm1 > 0, "astrometry context: em (Sun distance, au) and bm1 must be positive");
            for (int i = 0; i < 31; ++i) FV_REQUIRE(std::isfinite(reinterpret_cast<const double *>(&a)[i]), "astrometry context: not finite");
        }
    }
    void set_times(int ntimes, const double *rot) override {
        mhist_log.clear();  // entries index the previous configuration's time axis
        ntimes_topo = 0;
        astroms.clear();
        rots.resize(ntimes);
        for (int i = 0; i < ntimes; ++i) std::memcpy(rots[i].m, rot + 9 * i, 9 * sizeof(double));
    }
    void set_topo(int ntimes, int64_t n, const void *topo, int on_device) override {
        FV_HIP(hipSetDevice(device));
        FV_REQUIRE(n == nsrc, "topo source count != catalog (set_sources first)");
        mhist_log.clear();
        ntimes_topo = ntimes;
        astroms.clear();
        rots.assign(ntimes, Rot9{{1, 0, 0, 0, 1, 0, 0, 0, 1}});
        upload(d_topo, topo, sizeof(T) * 3 * (size_t)n * ntimes, on_device);
    }
    // (Re-setting what the handle already holds -- a caller that simulates the same array again and again -- leaves the
    // version of the target data alone: the column plans, unique-target runs and fused-gather records stay valid.)
    void set_freqs(int nf, const double *f) override {
        if ((size_t)nf == freqs.size() && nf > 0 && std::memcmp(freqs.data(), f, sizeof(double) * nf) == 0) return;
        ++targets_serial;  // tabulated target records (fused gather) are stale now
        FV_HIP(hipSetDevice(device));
        freqs.assign(f, f + nf);
        upload(d_freqs, f, sizeof(double) * nf, 0);
    }
    void set_array(const double *R, int64_t nb, const double *bls, int cop) override {
        if (!type1 && nbasis == 0 && nb == nbls && nb > 0 && coplanar == (cop != 0) && h_bls.size() == (size_t)3 * nb &&
            std::memcmp(rplane.m, R, 9 * sizeof(double)) == 0 && std::memcmp(h_bls.data(), bls, sizeof(double) * 3 * nb) == 0)
            return;  // the same array: the pair lists (and everything planned from them) stay
        ++targets_serial;  // tabulated target records (fused gather) are stale now
        FV_HIP(hipSetDevice(device));
        std::memcpy(rplane.m, R, 9 * sizeof(double));
        nbls = nb;
        nbasis = 0;
        type1 = false;
        coplanar = cop != 0;
        h_bls.assign(bls, bls + 3 * nb);
        std::vector<T> tmp(3 * nb);
        for (int64_t i = 0; i < 3 * nb; ++i) tmp[i] = (T)bls[i];
        upload(d_bls, tmp.data(), sizeof(T) * 3 * nb, 0);
        pairs.clear();
    }
    // Lattice array (cpu_simulate.py:661-681): integer baselines, n_modes = 2 max|bl| + 1 and the
    // basis matrix in seconds; topo is rotated by basis^T instead of the plane rotation (:964-965).
    void set_array_type1(const double *basis, int64_t nb, const int *bls_int, int n_modes) override {
        ++targets_serial;  // tabulated target records (fused gather) are stale now
        FV_HIP(hipSetDevice(device));
        FV_REQUIRE(n_modes >= 1 && n_modes % 2 == 1, "n_modes must be odd");
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) rplane.m[3 * i + j] = basis[3 * j + i];  // basis^T
        nbls = nb;
        nbasis = 0;
        coplanar = true;
        type1 = true;
        t1_nmodes = n_modes;
        h_bls.assign(3 * nb, 0.0);
        for (int64_t k = 0; k < nb; ++k) {
            FV_REQUIRE(std::abs(bls_int[k]) <= n_modes / 2 && std::abs(bls_int[nb + k]) <= n_modes / 2,
                       "integer baseline outside the mode range");
            h_bls[k] = bls_int[k];
            h_bls[nb + k] = bls_int[nb + k];
        }
        upload(d_blint, bls_int, sizeof(int) * 2 * nb, 0);
        pairs.clear();
    }
    // SURVEY App. B Q1 / Q2.  on (default): the reference's forms -- flipped baselines of a two-beam pair are
    // conjugated but their 2 x 2 block is not transposed (cpu_simulate.py:298); the eigenbeam (l, k) term reuses
    // V_kl(b)^T (:464-468, exact for real basis beams only).  off: V_ji(b) = V_ij(-b)^H and
    // V_lk(b) = conj(V_kl(-b))^T -- the transform is evaluated at -b as well.
    bool reference_compat = true;
    void set_reference_compat(int on) override { reference_compat = on != 0; }
    void set_nbeams(int n) override {
        beams.clear();
        beams.resize(n);
    }
    void set_beam_airy(int b, double diameter, const double *jones_scale, double power_scale) override {
        FV_REQUIRE(b >= 0 && b < (int)beams.size(), "beam index out of range");
        Beam &bm = beams[b];
        bm.kind = 0;
        bm.diameter = diameter;
        bm.ps = power_scale;
        bm.real_valued = true;
        for (int i = 0; i < 8; ++i) {
            bm.js[i] = jones_scale ? jones_scale[i] : (i % 2 ? 0.0 : 1.0);
            FV_REQUIRE(bm.js[i] == bm.js[i], "NaN in the Airy Jones factors");
            if (i % 2 && bm.js[i] != 0.0) bm.real_valued = false;
        }
    }
    void set_beam_table(int b, int nft, int nza, int naz, double za_max, const void *table,
                        int order) override {
        FV_HIP(hipSetDevice(device));
        FV_REQUIRE(b >= 0 && b < (int)beams.size(), "beam index out of range");
        FV_REQUIRE(nza >= 2 && naz >= 1 && nft >= 1 && za_max > 0, "bad beam table shape");
        FV_REQUIRE(order >= 0 && order <= 5, "beam interpolation order must be 0 .. 5");
        for (size_t i = 0; i < beams.size(); ++i)  // one spline_opts per simulation (cpu_simulate.py:557)
            FV_REQUIRE((int)i == b || beams[i].kind != 1 || beam_order == order,
                       "all tabulated beams of a handle share one interpolation order");
        beam_order = order;
        Beam &bm = beams[b];
        bm.kind = 1;
        bm.real_valued = true;
        if (polarized) {  // Jones tables with no imaginary part anywhere make every coherency product real
            const double *tv = static_cast<const double *>(table);
            const size_t nc = (size_t)nft * 4 * nza * naz;
            for (size_t i = 0; i < nc && bm.real_valued; ++i) bm.real_valued = tv[2 * i + 1] == 0.0;
        }
        bm.nfreq_tab = nft;
        bm.nza = nza;
        bm.naz = naz;
        bm.za_max = za_max;
        bm.table.reset(new DevBuf());
        const size_t per = polarized ? 4 * 16 : 8;  // complex128 Jones or float64 power
        if (!polarized) {
            upload(*bm.table, table, per * (size_t)nft * nza * naz, 0);
        } else {  // Jones tables live interleaved on the device (see eval_jones)
            DevBuf tmp;
            upload(tmp, table, per * (size_t)nft * nza * naz, 0);
            bm.table->reserve(per * (size_t)nft * nza * naz);
            const int64_t nodes = (int64_t)nza * naz;
            hipLaunchKernelGGL(k_jones_interleave, dim3((unsigned)cdiv(nodes * nft, 256)), dim3(256), 0, stream,
                               tmp.as<cplx<double>>(), bm.table->template as<cplx<double>>(), nodes, (int64_t)nft);
            FV_HIP(hipStreamSynchronize(stream));  // tmp goes out of scope
        }
        bspline_prefilter(bm.table->template as<double>(), nft, nza, naz, polarized ? 8 : 1, order, stream);
    }
    std::vector<int> in_bi, in_bj, in_idx;  // the caller's last pair lists, as given
    std::vector<int64_t> in_off;
    std::vector<signed char> in_fl;
    bool in_ordered = true;
    void set_beam_pairs(int np, const int *bi, const int *bj, const int64_t *off, const int *idx,
                        const signed char *flipped) override {
        {
            const int64_t tot = np > 0 ? off[np] : 0;
            if (!pairs.empty() && (int)pairs.size() == np && in_ordered == order_pairs && (int)in_bi.size() == np &&
                (int64_t)in_idx.size() == tot && std::equal(bi, bi + np, in_bi.begin()) && std::equal(bj, bj + np, in_bj.begin()) &&
                std::equal(off, off + np + 1, in_off.begin()) && std::equal(idx, idx + tot, in_idx.begin()) &&
                std::equal(flipped, flipped + tot, in_fl.begin()))
                return;  // the same lists on the same array (set_array clears the pairs when the array changes)
            in_bi.assign(bi, bi + np);
            in_bj.assign(bj, bj + np);
            in_off.assign(off, off + np + 1);
            in_idx.assign(idx, idx + tot);
            in_fl.assign(flipped, flipped + tot);
            in_ordered = order_pairs;
        }
        ++targets_serial;
        FV_HIP(hipSetDevice(device));
        FV_REQUIRE(nbls > 0 || np == 0, "set_array first");
        pairs.clear();
        for (int p = 0; p < np; ++p) {
            Pair pr;
            pr.bi = bi[p];
            pr.bj = bj[p];
            pr.n = off[p + 1] - off[p];
            const int *ix = idx + off[p];
            const signed char *fl = flipped + off[p];
            for (int64_t k = 0; k < pr.n; ++k) FV_REQUIRE(ix[k] >= 0 && ix[k] < nbls, "baseline index out of range");
            // The list is visited in order of the (flipped) baseline vector: redundant baselines -- most of a regular
            // array's -- then sit next to each other, the gather hands neighbouring items to one XCD, and they read
            // their common grid lines through one L2 (C3: 0.31 -> 0.25 ms per launch).  The order of a list is free:
            // every baseline writes its own output slot.
            static const bool keep_order = std::getenv("FFTVIS_HIP_NO_TARGET_SORT") != nullptr;
            pr.trivial0 = pr.n == nbls;
            for (int64_t k = 0; k < pr.n && pr.trivial0; ++k) pr.trivial0 = ix[k] == k && !fl[k];
            if (!pr.trivial0 && pr.n && type1) {
                pr.idx0.reset(new DevBuf());
                pr.flip0.reset(new DevBuf());
                upload(*pr.idx0, ix, sizeof(int) * pr.n, 0);
                upload(*pr.flip0, fl, pr.n, 0);
            }
            std::vector<int> six(ix, ix + pr.n);
            std::vector<signed char> sfl(fl, fl + pr.n);
            if (!keep_order && order_pairs && pr.n > 1) {
                double bmax = 0;
                for (int64_t k = 0; k < pr.n; ++k)
                    bmax = std::max({bmax, std::fabs(h_bls[ix[k]]), std::fabs(h_bls[(size_t)nbls + ix[k]])});
                const double q = 1e-7 * std::max(bmax, 1e-300);  // ties for vectors equal up to rounding
                std::vector<std::pair<int64_t, int64_t>> key(pr.n);
                std::vector<int64_t> ord(pr.n);
                for (int64_t k = 0; k < pr.n; ++k) {
                    const double sg = fl[k] ? -1.0 : 1.0;
                    key[k] = {(int64_t)std::llround(sg * h_bls[ix[k]] / q), (int64_t)std::llround(sg * h_bls[(size_t)nbls + ix[k]] / q)};
                    ord[k] = k;
                }
                std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });
                for (int64_t k = 0; k < pr.n; ++k) {
                    six[k] = ix[ord[k]];
                    sfl[k] = fl[ord[k]];
                }
            }
            ix = six.data();
            fl = sfl.data();
            pr.trivial = pr.n == nbls;
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int64_t k = 0; k < pr.n; ++k) {
                if (ix[k] != k || fl[k]) pr.trivial = false;
                const double sg = fl[k] ? -1.0 : 1.0;
                for (int d = 0; d < 3; ++d) {
                    const double v = sg * h_bls[(size_t)d * nbls + ix[k]];
                    lo[d] = std::min(lo[d], v);
                    hi[d] = std::max(hi[d], v);
                }
            }
            for (int d = 0; d < 3; ++d) {
                pr.btc[d] = pr.n ? 0.5 * (lo[d] + hi[d]) : 0.0;
                pr.B[d] = pr.n ? 0.5 * (hi[d] - lo[d]) * (1.0 + 1e-12) : 0.0;
                pr.Bs[d] = pr.n ? std::max(std::fabs(lo[d]), std::fabs(hi[d])) * (1.0 + 1e-12) : 0.0;
            }
            if (!pr.trivial && pr.n) {
                pr.idx.reset(new DevBuf());
                pr.flip.reset(new DevBuf());
                upload(*pr.idx, ix, sizeof(int) * pr.n, 0);
                upload(*pr.flip, fl, pr.n, 0);
            }
            pr.sorted = !keep_order && order_pairs && pr.n > 1;
            pr.h_idx = std::move(six);
            pr.h_flip = std::move(sfl);
            pairs.push_back(std::move(pr));
        }
    }

    // Redundant baselines are one target.  A regular array repeats most of its baseline vectors (HERA-350: 61 075
    // baselines, 7 957 distinct vectors), the visibility of a (beam pair, baseline vector) does not depend on WHICH
    // antennas form it, and the list is already visited in (u, v) order: runs of entries whose sign-adjusted vectors
    // agree to `tol` seconds in every component -- tol = what moves the phase 2 pi nu b . x by at most 1e-3 eps at the
    // run's highest frequency, i.e. far below the transform's own error; exact duplicates always qualify -- are gathered
    // once (k_interp walks the run for the output slots).  Compared with the run's FIRST entry, so runs cannot drift.
    // FFTVIS_HIP_NO_TARGET_DEDUP=1 turns it off.
    void build_unique(Pair &p, double tol) {
        const bool off = std::getenv("FFTVIS_HIP_NO_TARGET_DEDUP") != nullptr;  // read per run: tests flip it
        if (off) tol = -2.0;
        const int nd = wt_K ? 2 : 3;  // height terms: a run shares (u, v) only, every member brings its own b_z
        if (p.utol == tol && p.udims == nd) return;
        p.utol = tol;
        p.udims = nd;
        p.ustart.reset();
        p.upairs.reset();
        p.upairs_herm = -1;  // (pair_mirror_runs starts over on the new runs)
        p.h_ustart.clear();
        p.nu = p.nitems = p.n;
        ++targets_serial;  // column plans were built from the old runs
        if (off || !p.sorted) return;
        std::vector<int> st(1, 0);
        auto comp = [&](int64_t k, int d) { return (p.h_flip[k] ? -1.0 : 1.0) * h_bls[(size_t)d * nbls + p.h_idx[k]]; };
        for (int64_t k = 1; k < p.n; ++k) {
            const int64_t k0 = st.back();
            bool same = true;
            for (int d = 0; d < nd && same; ++d) same = std::fabs(comp(k, d) - comp(k0, d)) <= tol;
            if (!same) st.push_back((int)k);
        }
        st.push_back((int)p.n);
        const int64_t nu = (int64_t)st.size() - 1;
        if (nu * 10 > p.n * 9) return;  // (almost) nothing repeats: the plain list
        p.ustart.reset(new DevBuf());
        upload(*p.ustart, st.data(), sizeof(int) * st.size(), 0);
        p.nu = nu;
        p.h_ustart = std::move(st);
    }

    // Packed runs gather every target at s and at -s: the run of baselines b and the run of baselines -b want the same
    // two evaluations and share one item (k_interp: upairs).  Runs are matched through a hash of their vectors rounded
    // to 4 tol (the 27 neighbouring cells are searched: either vector may sit next to a rounding boundary).
    void pair_mirror_runs(Pair &p, double tol) {
        const bool want = p.herm != 0 && p.ustart && tol > 0.0 && !std::getenv("FFTVIS_HIP_NO_TARGET_PAIRS");
        const int nd = p.udims;
        if (p.upairs_herm == (want ? nd : 0)) return;
        p.upairs_herm = want ? nd : 0;
        p.upairs.reset();
        p.nitems = p.nu;
        ++targets_serial;
        if (!want) return;
        const double q = 4.0 * tol;
        struct Key {
            int64_t a, b, c;
            bool operator==(const Key &o) const { return a == o.a && b == o.b && c == o.c; }
        };
        struct Hash {
            size_t operator()(const Key &k) const { return (size_t)(k.a * 0x9E3779B97F4A7C15ull) ^ (size_t)(k.b * 0xC2B2AE3D27D4EB4Full) ^ (size_t)(k.c * 0x165667B19E3779F9ull); }
        };
        auto vec = [&](int64_t u, int d) {
            const int64_t kl = p.h_ustart[u];
            return (p.h_flip[kl] ? -1.0 : 1.0) * h_bls[(size_t)d * nbls + p.h_idx[kl]];
        };
        std::unordered_map<Key, int, Hash> at;
        at.reserve((size_t)p.nu * 2);
        auto cell2 = [&](double v) { return nd > 2 ? (int64_t)std::llround(v / q) : (int64_t)0; };
        for (int64_t u = 0; u < p.nu; ++u) at[Key{(int64_t)std::llround(vec(u, 0) / q), (int64_t)std::llround(vec(u, 1) / q), cell2(vec(u, 2))}] = (int)u;
        std::vector<int> partner((size_t)p.nu, -1), items;
        for (int64_t u = 0; u < p.nu; ++u) {
            if (partner[u] >= 0) continue;
            const int64_t k0 = std::llround(-vec(u, 0) / q), k1 = std::llround(-vec(u, 1) / q), k2 = cell2(-vec(u, 2));
            int best = -1;
            for (int da = -1; da <= 1 && best < 0; ++da)
                for (int db = -1; db <= 1 && best < 0; ++db)
                    for (int dc = (nd > 2 ? -1 : 0); dc <= (nd > 2 ? 1 : 0) && best < 0; ++dc) {
                        auto it = at.find(Key{k0 + da, k1 + db, k2 + dc});
                        if (it == at.end()) continue;
                        const int v = it->second;
                        if (v == (int)u || partner[v] >= 0) continue;
                        bool same = true;
                        for (int d = 0; d < nd && same; ++d) same = std::fabs(vec(v, d) + vec(u, d)) <= tol;
                        if (same) best = v;
                    }
            if (best >= 0) {
                partner[u] = best;
                partner[best] = (int)u;
            }
        }
        for (int64_t u = 0; u < p.nu; ++u) {
            if (partner[u] >= 0 && partner[u] < (int)u) continue;  // listed with its partner
            items.push_back((int)u);
            items.push_back(partner[u]);
        }
        if ((int64_t)items.size() / 2 == p.nu) return;  // no mirror pairs
        p.upairs.reset(new DevBuf());
        upload(*p.upairs, items.data(), sizeof(int) * items.size(), 0);
        p.nitems = (int64_t)items.size() / 2;
    }

    // Column plan (Nufft3::arm_columns): which columns of the transform's first dimension the targets of one (frequency
    // group, beam pair) read at all, per frequency -- the footprints of the distinct target vectors (and of their
    // mirror images where the run gathers at -s too), exactly as k_interp places them; a footprint whose first column
    // is within 1e-6 of a rounding boundary takes both candidates.  Compact numbers follow the residue-major
    // position order, so that a residue job of the x-pass stores runs of neighbouring compact columns.  Host
    // arithmetic, once per (targets, group geometry); kept for later runs.  Not used when it would keep more than
    // 85 % of the columns (arrays without repeated baseline vectors).  FFTVIS_HIP_NO_COLUMN_PLAN=1 turns it off.
    struct ColPlan {
        int64_t serial;
        int pair, fa, fb, nos, sP, cnt, n2, no, w;
        double h, btc;
        bool both;
        DevBuf tab, xtab;  // the gather's table (1 + compact column) and the x-pass's (1 + element index of that column)
        int yna = 0, blk = 0;
        int ncc = 0;
        bool use = false;
        // y-pass output mask (k_plan_rowmask): which 16-output chunks of each (column block, residue) hold a footprint row
        int yn2 = 0, yno = 0, yP = 0, yQ = 0;
        double yh = 0, ybtc = 0;
        DevBuf omask;
        int nblk = 0;
        double out_cells = 0;  // cells of C per transform the masked y-pass stores
    };
    std::vector<std::unique_ptr<ColPlan>> col_plans;
    std::vector<ColPlan *> col_plan_of;  // [group * pairs + pair] of the current run
    std::vector<ColPlan *> col_plan_of_l[2];  // the same for the light height terms' plans (wt_k0, wt_k1)
    ColPlan *column_plan(int pi, const Pair &pr, int fa, int fb, Nufft3<T> *n0) {
        const DimGeom &x = n0->geo.d[0], &y = n0->geo.d[1];
        const int w = n0->ker.w;
        const bool both = pr.herm || pr.mirror;
        for (auto &c : col_plans)
            if (c->serial == targets_serial && c->pair == pi && c->fa == fa && c->fb == fb && c->nos == x.nos() &&
                c->sP == x.sP() && c->cnt == x.cnt() && c->n2 == x.n2 && c->no == x.no && c->w == w && c->h == x.h &&
                c->btc == x.btc && c->both == both && c->yn2 == y.n2 && c->yno == y.no && c->yP == y.P && c->yQ == y.Q &&
                c->yh == y.h && c->ybtc == y.btc && c->yna == y.na && c->blk == n0->b_block_log_public())
                return c.get();
        std::unique_ptr<ColPlan> c(new ColPlan{targets_serial, pi, fa, fb, x.nos(), x.sP(), x.cnt(), x.n2, x.no, w, x.h, x.btc, both});
        const int nfg = fb - fa, stride = x.nos(), P = x.sP(), cnt = x.cnt();
        std::vector<int> tab((size_t)nfg * stride, 0);
        const int64_t nu = pr.h_ustart.empty() ? pr.n : (int64_t)pr.h_ustart.size() - 1;
        c->use = true;
        for (int fg = 0; fg < nfg && c->use; ++fg) {
            int *row = tab.data() + (size_t)fg * stride;
            const double sc = freqs[fa + fg];
            for (int64_t ui = 0; ui < nu; ++ui) {
                const int64_t kl = pr.h_ustart.empty() ? ui : pr.h_ustart[ui];
                const int64_t k = pr.h_idx[kl];
                const double sg = pr.h_flip[kl] ? -1.0 : 1.0;
                const double sv = sc * sg * (double)(T)h_bls[k];  // as k_interp forms it from the device copy
                const double th = x.h * (sv - sc * x.btc);
                for (int side = 0; side < (both ? 2 : 1); ++side) {
                    const double e = (side ? -1.0 : 1.0) * th * x.n2 * (0.5 / M_PI) + 0.5 * x.no;
                    const double t = e - 0.5 * w, jc = std::ceil(t);
                    const bool amb = jc - t < 1e-6 || jc - t > 1.0 - 1e-6;
                    int lo = std::max(0, std::min(x.no - w, (int)jc)), hi = lo + w - 1;
                    if (amb) {
                        lo = std::max(0, lo - 1);
                        hi = std::min(x.no - 1, hi + 1);
                    }
                    for (int i = lo; i <= hi; ++i) row[out_pos(i, P, cnt)] = 1;
                }
            }
            int run = 0;
            for (int i = 0; i < stride; ++i)
                if (row[i]) row[i] = ++run;
            c->ncc = std::max(c->ncc, run);
            if (run * 100 > x.no * 85) c->use = false;  // nothing to gain (decided on the first frequency already)
        }
        c->ncc = (c->ncc + 7) / 8 * 8;
        c->yn2 = y.n2; c->yno = y.no; c->yP = y.P; c->yQ = y.Q; c->yh = y.h; c->ybtc = y.btc;
        c->yna = y.na;
        c->blk = n0->b_block_log_public();
        if (c->use) {
            upload(c->tab, tab.data(), sizeof(int) * tab.size(), 0);
            // the x-pass stores compact column cc of a row at element (cc >> b) (na_y << b) + (cc & (2^b - 1)) of the row's
            // blocked output (RowDifArgs::out_blk): tabulated, so that a store index is one subtraction
            {
                std::vector<int> xt(tab.size());
                const int b = c->blk, rows = y.na << b, mask = (1 << b) - 1;
                for (size_t i = 0; i < tab.size(); ++i) xt[i] = tab[i] ? ((tab[i] - 1) >> b) * rows + ((tab[i] - 1) & mask) + 1 : 0;
                upload(c->xtab, xt.data(), sizeof(int) * xt.size(), 0);
            }
            // the y-pass output mask, on the device from the same targets
            if (y.logQ >= 9 && y.logQ <= 11 && !std::getenv("FFTVIS_HIP_NO_OUTPUT_MASK")) {
                const int bl = n0->ypass_cols_log(), nw = y.Q > 1024 ? y.Q / 1024 : 1;
                c->nblk = (c->ncc + (1 << bl) - 1) >> bl;
                const size_t words = (size_t)nfg * c->nblk * y.P * nw;
                c->omask.reserve(sizeof(unsigned long long) * words);
                FV_HIP(hipMemsetAsync(c->omask.p, 0, sizeof(unsigned long long) * words, stream));
                hipStream_t keep = n0->stream;
                n0->stream = stream;
                n0->build_rowmask(nu, d_bls.as<T>(), d_bls.as<T>() + nbls, pr.trivial ? nullptr : pr.idx->template as<int>(),
                                  pr.trivial ? nullptr : pr.flip->template as<signed char>(),
                                  pr.ustart ? pr.ustart->template as<int>() : nullptr, d_freqs.as<double>() + fa, nfg, both,
                                  c->tab.template as<int>(), c->ncc, c->omask.template as<unsigned long long>(), c->nblk, nw);
                n0->stream = keep;
                std::vector<unsigned long long> hm(words);
                FV_HIP(hipMemcpyAsync(hm.data(), c->omask.p, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost, stream));
                FV_HIP(hipStreamSynchronize(stream));
                double bits = 0;
                for (unsigned long long v : hm) bits += __builtin_popcountll(v);
                c->out_cells = bits * 16.0 * (1 << bl) / nfg;
            }
        }
        col_plans.push_back(std::move(c));
        return col_plans.back().get();
    }

    // Eigenbeam mode (cpu_simulate.py:303-470): beams 0..K-1 are basis beams; every (k <= l) term
    // runs over ALL baselines without flips (:402-404) and is contracted with the coefficients.
    void set_basis(int nant, int K, int nfreq, const void *coefs, const int *ant1,
                   const int *ant2) override {
        FV_HIP(hipSetDevice(device));
        FV_REQUIRE(polarized, "basis beams need a polarized engine (wrapper.py:280-283)");
        FV_REQUIRE(nbls > 0 && K >= 1 && K == (int)beams.size(), "set_array and the K basis beams first");
        FV_REQUIRE(nfreq == (int)freqs.size(), "beam_coefs frequency axis != freqs");
        for (int64_t b = 0; b < nbls; ++b)
            FV_REQUIRE(ant1[b] >= 0 && ant1[b] < nant && ant2[b] >= 0 && ant2[b] < nant, "antenna index out of range");
        nbasis = K;
        upload(d_coefs, coefs, sizeof(cplx<T>) * (size_t)nant * K * nfreq, 0);
        upload(d_ant1, ant1, sizeof(int) * nbls, 0);
        upload(d_ant2, ant2, sizeof(int) * nbls, 0);
        std::vector<int> bi, bj, idx(nbls);
        std::vector<int64_t> off(1, 0);
        std::vector<int> all;
        std::vector<signed char> fl;
        for (int k = 0; k < K; ++k)
            for (int l = k; l < K; ++l) {
                bi.push_back(k);
                bj.push_back(l);
                for (int64_t b = 0; b < nbls; ++b) {
                    all.push_back((int)b);
                    fl.push_back(0);
                }
                off.push_back((int64_t)all.size());
            }
        // (u, v) order here too: redundant baselines are gathered once (build_unique) and only the contraction with the
        // per-antenna coefficients runs per baseline
        set_beam_pairs((int)bi.size(), bi.data(), bj.data(), off.data(), all.data(), fl.data());
    }

    void set_chunking(int nchunks, double sb) override {
        FV_REQUIRE(nchunks >= 1, "nchunks must be >= 1");
        FV_REQUIRE(sb > 0.0 && sb <= 1.0, "source_buffer must be in (0, 1]");
        src_chunks = nchunks;
        source_buffer = sb;
    }

    // rotate -> horizon cut -> az/za -> 2 pi R topo for time ti; returns the device address of the
    // live above-horizon count (it never visits the host inside the loop).
    // The step works on the catalog range [s0, s0 + sn) (one source chunk); cap = capacity of the
    // compacted arrays; hslot = where the live count is kept for stats().
    const int *horizon_step(Lane &L, int ti, int64_t cap, int nblk, hipStream_t on, int64_t s0, int64_t sn,
                            int64_t hslot) {
        hipStream_t stream = on ? on : L.stream;
        DevBuf &d_blockcnt = L.d_blockcnt, &d_blockoff = L.d_blockoff, &d_scan_tot = L.d_scan_tot,
               &d_scan_off = L.d_scan_off, &d_xyz = L.d_xyz, &d_az = L.d_az, &d_za = L.d_za,
               &d_srcidx = L.d_srcidx;
        // R_t . eq on the fly, topocentric vectors the caller computed, or this time's astrometry context applied to the
        // chunk's sources first (into the lane's own (3, nsrc) scratch)
        const T *vec = ntimes_topo ? d_topo.as<T>() + (size_t)ti * 3 * nsrc : d_eq.as<T>();
        if (!astroms.empty()) {
            L.d_enu.reserve(sizeof(T) * 3 * (size_t)std::max<int64_t>(nsrc, 1));
            hipLaunchKernelGGL(k_astrom_topo<T>, dim3((unsigned)cdiv(sn, 256)), dim3(256), 0, stream, sn, nsrc, s0,
                               d_eq.as<T>(), astroms[ti], L.d_enu.template as<T>());
            vec = L.d_enu.template as<T>();
        }
        hipLaunchKernelGGL(k_horizon_count<T>, dim3(nblk), dim3(256), 0, stream, sn, nsrc, s0, vec,
                           rots[ti], d_blockcnt.as<int>());
        if (nblk <= 4096) {
            hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, stream,
                               d_blockcnt.as<int>(), d_blockoff.as<int>(), nblk);
        } else {
            const int nb2 = (int)cdiv(nblk, 1024);
            d_scan_tot.reserve(sizeof(int) * (nb2 + 1));
            d_scan_off.reserve(sizeof(int) * (nb2 + 1));
            hipLaunchKernelGGL(k_scan_blocks, dim3(nb2), dim3(1024), 0, stream,
                               d_blockcnt.as<int>(), d_blockoff.as<int>(), d_scan_tot.as<int>(), nblk);
            hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, stream,
                               d_scan_tot.as<int>(), d_scan_off.as<int>(), nb2);
            hipLaunchKernelGGL(k_scan_add, dim3(nb2), dim3(1024), 0, stream,
                               d_blockoff.as<int>(), d_scan_off.as<int>(), nblk);
        }
        hipLaunchKernelGGL(k_horizon_compact<T>, dim3(nblk), dim3(256), 0, stream, sn, nsrc, s0, vec,
                           rots[ti], rplane, d_blockoff.as<int>(), d_xyz.as<T>(), cap,
                           d_az.as<T>(), d_za.as<T>(), d_srcidx.as<int>(), d_err.as<int>() + 2);
        const int *Mp = d_blockoff.as<int>() + nblk;
        FV_HIP(hipMemcpyAsync(d_mhist.as<int>() + hslot, Mp, sizeof(int), hipMemcpyDeviceToDevice, stream));
        return Mp;
    }

    // ---- type-1 run: per time, per frequency batch: bin (source, freq) entries on periodic
    // n2 x n2 planes, strengths, gather-spread, pruned FFT to the n_modes central modes, pick.
    void run_type1(int t0, int t1, int f0, int f1, void *out, int out_on_device) {
        const int nt = t1 - t0, nf = f1 - f0;
        const int64_t per_tf = (int64_t)tpol * nbls;
        const size_t out_bytes = sizeof(cplx<T>) * (size_t)nf * nt * per_tf;
        cplx<T> *dout;
        if (out_on_device) {
            dout = (cplx<T> *)out;
        } else {
            d_out.reserve(std::max<size_t>(out_bytes, 16));
            dout = d_out.as<cplx<T>>();
        }
        FV_HIP(hipMemsetAsync(dout, 0, out_bytes, stream));
        // host output: the lattice path computes a C3 block in 65 ms, so the 10-GB copy IS the call -- the caller's array
        // is touched in parallel and pinned while the kernels run (HostPin), then one copy at PCIe rate (0.72 -> 0.3 s)
        const bool drain = !out_on_device && out_bytes >= drain_min_bytes();
        HostPin pin;
        pin.drain_on = &stream;  // copy_block_pinned below rides on the main stream
        // destination layout of this run (fv_sim_run_into), consumed here
        const int64_t out_fs = out_f_stride ? out_f_stride : (int64_t)nt * per_tf;
        const bool shared = out_shared != 0;
        out_f_stride = 0;
        out_shared = 0;
        FV_REQUIRE(out_fs >= (int64_t)nt * per_tf, "fv_sim_run_into: the channel stride is shorter than a channel's run");
        const size_t run_bytes = sizeof(cplx<T>) * (size_t)nt * per_tf;
        // (a block inside a larger array is pinned run by run: only worth it -- and only safe against two runs meeting in
        // one page -- when the runs are long)
        const bool pinnable = out_fs == (int64_t)nt * per_tf || (run_bytes >= ((size_t)1 << 20) && (size_t)(out_fs - (int64_t)nt * per_tf) * sizeof(cplx<T>) >= 8192);
        if (drain && pinnable) pin.start(device, out, (size_t)nf, run_bytes, sizeof(cplx<T>) * (size_t)out_fs, shared);
        const double sigma = this->sigma == 0.0 ? 2.0 : this->sigma;  // "auto" is a type-3 matter
        sigma_run = sigma;
        if (!t1fft) t1fft.reset(new Nufft3<T>(2, eps, sigma, stream));
        const KerParams &ker = t1fft->ker;
        // grid: n2 = P Q >= sigma n_modes (and >= 2 w), all n2 inputs live, n_modes + 1 outputs kept
        DimGeom g;
        g.n1 = t1_nmodes;
        choose_pq(std::max((int)std::ceil(sigma * t1_nmodes), 2 * ker.w + 16), g);
        g.na = g.n2;
        g.no = t1_nmodes + 1;
        t1fft->set_fft_geometry(g, g);
        t1_dec.reserve(sizeof(T) * g.no);
        hipLaunchKernelGGL(k_deconv_table<T>, dim3(cdiv(g.no, 256)), dim3(256), 0, stream, g.no,
                           g.n2, ker, t1_dec.as<T>());
        const int nb1 = (g.n2 + T1_PAD) >> BINLOG;
        int64_t pol_off[4] = {0, 0, 0, 0};
        if (polarized)
            for (int r = 0; r < 4; ++r) pol_off[r] = (int64_t)((r % 2) * 2 + r / 2) * nbls;

        const int nch = (int)std::max<int64_t>(1, std::min<int64_t>(src_chunks, nsrc));
        const int64_t csz = std::max<int64_t>(cdiv(nsrc, nch), 1);
        const int64_t cap = std::max<int64_t>((int64_t)std::ceil(csz * source_buffer), 1);
        const int nblk = (int)cdiv(csz, 256);
        for (Lane &Lr : lanes) {
            Lr.d_xyz.reserve(sizeof(T) * 3 * cap);
            Lr.d_az.reserve(sizeof(T) * cap);
            Lr.d_za.reserve(sizeof(T) * cap);
            Lr.d_srcidx.reserve(sizeof(int) * cap);
            Lr.d_blockcnt.reserve(sizeof(int) * (nblk + 1));
            Lr.d_blockoff.reserve(sizeof(int) * (nblk + 1));
        }
        reserve_mhist(sizeof(int) * rots.size() * std::max(1, src_chunks));
        // frequencies per batch: bounded by entries (~1.3 per (source, freq)) and by grid bytes
        const char *eb = std::getenv("FFTVIS_HIP_GRID_BYTES");
        const double budget = eb ? std::atof(eb) : 8.0 * 1024 * 1024 * 1024;
        const double plane_bytes = 2.0 * g.n2 * (double)g.n2 * tpol * sizeof(cplx<T>);
        // entries per live (source, frequency) pair: 1 + the periodic images of footprints that cross an
        // edge of the n2 x n2 plane -- (1 + (w + 1) / n2)^2 on average for uniformly placed sources (1.52
        // at w = 16 on the smallest, 64-cell planes); 5 % head-room on top, and a catalog that still
        // overflows (sources piled on a plane edge) fails the run (t1 overflow flag), never silently
        const double img = (1.0 + (ker.w + 1.0) / g.n2) * (1.0 + (ker.w + 1.0) / g.n2) * 1.05;
        int nfb = (int)std::max(1.0, std::min({(double)nf, budget / plane_bytes, 24.0e6 / (img * cap)}));
        const int64_t ecap = (int64_t)(img * cap * nfb) + 4096;
        const int nbins = nfb * nb1 * nb1;
        const int rec = t1_record_bytes(ker.w, sizeof(T));
        // Pipelined like the type-3 loop: the entry sort of unit (time, batch) u+1 (three kernels over
        // every (source, frequency) pair, ~25 % of a step) runs on the low-priority stream beside the
        // strengths / spread / FFT / pick of unit u; two sets of sort buffers and two sets of
        // per-time source arrays alternate.
        const char *ep = std::getenv("FFTVIS_HIP_PIPE");
        const int nunits = nt * nch * (int)cdiv(nf, nfb);
        const bool pipe = timing_level != 2 && nunits > 1 && !(ep && std::atoi(ep) == 0);
        const hipStream_t ps = pipe ? prep_stream : stream;
        for (int sset = 0; sset < (pipe ? 2 : 1); ++sset) {
            t1_meta[sset].reserve(sizeof(int) * (2 * (size_t)(nbins + 1) + 2));
            t1_binstart[sset].reserve(sizeof(int) * (nbins + 1));
            t1_rec[sset].reserve((size_t)rec * ecap);
        }
        t1_cs.reserve(sizeof(cplx<T>) * ecap * tpol);
        bool set_pending[2] = {false, false}, lane_pending[2] = {false, false};
        lane_mode = -1;  // a type-3 run after this one drains the streams before it reuses the lanes
        if (pipe) {  // the sort may start once the set-up queued on the main stream is done
            FV_HIP(hipEventRecord(ev_start, stream));
            FV_HIP(hipStreamWaitEvent(ps, ev_start, 0));
        }
        int unit = 0;

        for (int tc = 0; tc < nt * nch; ++tc) {  // (time, source chunk), chunks innermost (cpu_simulate.py:936-939)
            const int ti = t0 + tc / nch, chunk = tc % nch;
            const int64_t s0 = (int64_t)chunk * csz, sn = std::min<int64_t>(csz, nsrc - s0);
            if (nsrc == 0 || sn <= 0) continue;
            const int li = pipe ? tc % 2 : 0;
            Lane &L = lanes[li];
            DevBuf &d_xyz = L.d_xyz, &d_az = L.d_az, &d_za = L.d_za, &d_srcidx = L.d_srcidx;
            if (pipe && lane_pending[li]) FV_HIP(hipStreamWaitEvent(ps, L.done, 0));  // its strengths are done
            size_t e0 = ev_begin(TM_PREP, ps);
            const int *Mp = horizon_step(L, ti, cap, nblk, ps, s0, sn, (int64_t)ti * nch + chunk);
            ev_end(e0, ps);
            mhist_log.push_back({ti * nch + chunk, 0.0});
            const size_t hist_slot = mhist_log.size() - 1;
            for (int fa = f0; fa < f1; fa += nfb, ++unit) {
                const int nfg = std::min(nfb, f1 - fa);
                const int ss = pipe ? unit % 2 : 0;
                DevBuf &meta = t1_meta[ss], &binstart = t1_binstart[ss], &recs = t1_rec[ss];
                T1Args a{};
                a.n2 = g.n2;
                a.nb1 = nb1;
                a.w = ker.w;
                a.nfg = nfg;
                a.f_first = fa;
                a.cap = cap;
                a.ecap = ecap;
                a.rec = rec;
                const int nbn = nfg * nb1 * nb1;
                int *counts_p = meta.as<int>(), *cursor_p = counts_p + (nbins + 1), *ovf_p = d_err.as<int>() + 1;
                if (pipe && set_pending[ss]) FV_HIP(hipStreamWaitEvent(ps, lanes[ss].heavy_done, 0));
                size_t e1 = ev_begin(TM_PREP, ps);
                FV_HIP(hipMemsetAsync(meta.p, 0, sizeof(int) * (2 * (size_t)(nbins + 1) + 2), ps));
                const dim3 gb((unsigned)cdiv(cap * nfg, 256));
                hipLaunchKernelGGL((k_t1_bin<T, true>), gb, dim3(256), 0, ps, a, Mp, d_xyz.as<T>(),
                                   d_freqs.as<double>(), counts_p, (const int *)nullptr, cursor_p,
                                   (unsigned char *)nullptr, (T)ker.beta, (T)ker.c, ovf_p);
                t1fft->stream = ps;
                t1fft->exclusive_scan(counts_p, binstart.as<int>(), nbn);
                t1fft->stream = stream;
                hipLaunchKernelGGL((ker.w == 9 ? k_t1_bin<T, false, 9> : ker.w == 5 ? k_t1_bin<T, false, 5> : ker.w == 7 ? k_t1_bin<T, false, 7> : k_t1_bin<T, false, 0>), gb, dim3(256), 0, ps, a, Mp,
                                   d_xyz.as<T>(), d_freqs.as<double>(), counts_p, (const int *)binstart.as<int>(),
                                   cursor_p, recs.as<unsigned char>(), (T)ker.beta, (T)ker.c, ovf_p);
                ev_end(e1, ps);
                if (pipe) {
                    FV_HIP(hipEventRecord(lanes[ss].prep_done, ps));
                    FV_HIP(hipStreamWaitEvent(stream, lanes[ss].prep_done, 0));
                }
                const int *nent = binstart.as<int>() + nbn;
                for (const Pair &pr : pairs) {
                    if (pr.n == 0) continue;
                    size_t e2 = ev_begin(TM_STRENGTHS, stream);
                    StrengthArgs sa{};
                    sa.M = cap;
                    sa.nfg = nfg;
                    sa.f_first = fa;
                    sa.nfreq = nfreq_cat;
                    sa.polarized = polarized;
                    sa.pol_sky = pol_sky;
                    sa.same_beam = pr.bi == pr.bj;
                    // Hermitian packing: two planes per frequency instead of four for a single-beam pair
                    const bool herm1 = polarized && pr.bi == pr.bj && std::getenv("FFTVIS_HIP_NO_HERMITIAN") == nullptr;
                    const int tg = herm1 ? 2 : tpol;
                    sa.herm = herm1;
                    sa.dim = 2;
                    sa.bi = desc(pr.bi);
                    sa.bj = desc(pr.bj);
                    hipLaunchKernelGGL((beam_order == 3 ? k_t1_strengths<T, 3> : beam_order == 1 ? k_t1_strengths<T, 1> : k_t1_strengths<T, 0>),
                                       dim3(cdiv(ecap, 256)), dim3(256), 0, stream,
                                       sa, nent, ecap, (const unsigned char *)recs.as<unsigned char>(), rec,
                                       d_srcidx.as<int>(),
                                       d_az.as<T>(), d_za.as<T>(), d_flux.p, d_freqs.as<double>(),
                                       t1_cs.as<cplx<T>>());
                    ev_end(e2, stream);
                    const int nplanes = nfg * tg;
                    cplx<T> *A = t1fft->fft_input(nplanes);
                    size_t e3 = ev_begin(TM_SPREAD, stream);
                    const dim3 gs((unsigned)cdiv(g.n2 >> (BINLOG + 1), 4), (unsigned)(g.n2 >> (BINLOG + 1)), (unsigned)nfg);  // 16 x 16 cells per wave
                    // fp64: the accumulation on the matrix pipe (FFTVIS_HIP_T1_MM=0: the vector version)
                    static const bool t1_mm = !(std::getenv("FFTVIS_HIP_T1_MM") && std::atoi(std::getenv("FFTVIS_HIP_T1_MM")) == 0);
                    bool spread_done = false;
                    if constexpr (sizeof(T) == 8) {
                        if (t1_mm) {
                            auto kmm = herm1 ? k_t1_spread_mm<2> : polarized ? k_t1_spread_mm<4> : k_t1_spread_mm<1>;
                            hipLaunchKernelGGL(kmm, gs, dim3(SPREAD_THREADS), 0, stream, a,
                                               (const unsigned char *)recs.as<unsigned char>(), (const int *)binstart.as<int>(),
                                               (const cplx<double> *)t1_cs.as<cplx<double>>(), (cplx<double> *)A);
                            spread_done = true;
                        }
                    }
                    if (spread_done) {
                    } else if (herm1)
                        hipLaunchKernelGGL((k_t1_spread<T, 2>), gs, dim3(SPREAD_THREADS), 0, stream, a,
                                           (const unsigned char *)recs.as<unsigned char>(),
                                           (const int *)binstart.as<int>(),
                                           (const cplx<T> *)t1_cs.as<cplx<T>>(), A);
                    else if (polarized)
                        hipLaunchKernelGGL((k_t1_spread<T, 4>), gs, dim3(SPREAD_THREADS), 0, stream, a,
                                           (const unsigned char *)recs.as<unsigned char>(),
                                           (const int *)binstart.as<int>(),
                                           (const cplx<T> *)t1_cs.as<cplx<T>>(), A);
                    else
                        hipLaunchKernelGGL((k_t1_spread<T, 1>), gs, dim3(SPREAD_THREADS), 0, stream, a,
                                           (const unsigned char *)recs.as<unsigned char>(),
                                           (const int *)binstart.as<int>(),
                                           (const cplx<T> *)t1_cs.as<cplx<T>>(), A);
                    ev_end(e3, stream);
                    st[0] += 1;
                    st[1] += (double)g.n2 * g.n2 * nplanes;
                    mhist_log[hist_slot].second += nplanes;
                    size_t e4 = ev_begin(TM_FFT, stream);
                    t1fft->fft(nplanes);
                    ev_end(e4, stream);
                    st[3] += ((double)g.n2 * g.n2 + 2.0 * g.no * g.n2 + (double)g.no * g.no) * nplanes;
                    size_t e5 = ev_begin(TM_INTERP, stream);
                    cplx<T> *obase = dout + ((int64_t)(fa - f0) * nt + (ti - t0)) * per_tf;
                    hipLaunchKernelGGL(k_t1_pick<T>, dim3(cdiv(pr.n * nfg, 256)), dim3(256), 0, stream,
                                       t1fft->fft_output(), g.no, g.P, g.cnt(), nfg, tg,
                                       (const int *)d_blint.as<int>(),
                                       (const int *)d_blint.as<int>() + nbls, pr.n,
                                       pr.trivial0 ? (const int *)nullptr : (const int *)(pr.idx0 ? pr.idx0 : pr.idx)->template as<int>(),
                                       pr.trivial0 ? (const signed char *)nullptr
                                                   : (const signed char *)(pr.flip0 ? pr.flip0 : pr.flip)->template as<signed char>(),
                                       (const T *)t1_dec.as<T>(), obase, (int64_t)nt * per_tf, pol_off[0],
                                       pol_off[1], pol_off[2], pol_off[3], chunk > 0, herm1, !reference_compat);
                    ev_end(e5, stream);
                    st[4] += (double)pr.n * nplanes;
                    st[6] = g.n2;
                    st[7] = g.n2;
                    st[8] = g.n2 * 65536.0 + g.n2;
                    st[9] = ker.w;
                }
                if (pipe) {
                    FV_HIP(hipEventRecord(lanes[ss].heavy_done, stream));
                    set_pending[ss] = true;
                }
            }
            if (pipe) {
                FV_HIP(hipEventRecord(L.done, stream));
                lane_pending[li] = true;
            }
        }
        if (!out_on_device) {
            copy_block_to_host(out, dout, nf, (int64_t)nt * per_tf, out_fs, drain && pinnable && pin.wait(), stream);
            FV_HIP(hipStreamSynchronize(stream));
            if (timing_level) ev_collect();
            check_errors();
        }
    }

    // Tight box of {2 pi R_plane v : |v| = 1, v_up >= 0} per coordinate.
    void source_box(double *xc, double *X) const {
        for (int d = 0; d < 3; ++d) {
            const double al = rplane.m[3 * d + 2];
            const double rad = std::sqrt(std::max(0.0, 1.0 - al * al));
            const double hi = al >= 0 ? 1.0 : rad, lo = al <= 0 ? -1.0 : -rad;
            xc[d] = 2.0 * M_PI * 0.5 * (hi + lo);
            X[d] = 2.0 * M_PI * 0.5 * (hi - lo) * (1.0 + 1e-9) + 1e-12;
        }
    }

    // Split [f0, f1) into groups of consecutive channels sharing one fine-grid geometry (sized
    // for the group's top frequency).  Small grids are launch-bound, so they tolerate a wide
    // frequency ratio (more wasted cells, far fewer launches); large grids are HBM-bound and get
    // a narrow one.  cells_top = fine-grid cells per transform at the highest frequency.
    std::vector<std::pair<int, int>> freq_groups(int f0, int f1, double cells_top, int tg) const {
        const char *er = std::getenv("FFTVIS_HIP_GROUP_RATIO");
        const char *eb = std::getenv("FFTVIS_HIP_GRID_BYTES");
        // grid bytes per launch, measured on C3: round 1 (four transforms per frequency) 3-4 GiB 350 ms per two
        // time steps, 8 GiB 362, 2 GiB 364, 1 GiB 369; round 2 (two per frequency, groups of whole eights):
        // 6 GiB 416.7 ms per four time steps (spread at 0.64 of the HBM roofline), 4 GiB 421.7 (0.54), 3 GiB 422.7
        const double budget = eb ? std::atof(eb) : 6.0 * 1024 * 1024 * 1024;
        double fmax = 1.0;
        for (int f = f0; f < f1; ++f) fmax = std::max(fmax, std::fabs(freqs[f]));
        const double mb = cells_top * sizeof(cplx<T>) / (1024.0 * 1024.0);
        double ratio = 0.5 + 0.4 * std::min(1.0, std::max(0.0, std::log2(mb / 16.0) / 4.0));
        if (er) ratio = std::atof(er);
        std::vector<std::pair<int, int>> g;
        int a = f0;
        while (a < f1) {
            double lo = std::fabs(freqs[a]), hi = lo;
            int b = a + 1;
            while (b < f1) {
                const double nlo = std::min(lo, std::fabs(freqs[b])), nhi = std::max(hi, std::fabs(freqs[b]));
                if (nlo < ratio * nhi) break;
                const double sc = nhi / fmax;
                const double bytes = cells_top * sc * sc * (b + 1 - a) * tg * sizeof(cplx<T>);
                if (bytes > budget) break;
                lo = nlo;
                hi = nhi;
                ++b;
            }
            // whole spread chunks: a launch's transforms run as kernel launches of 16 / 8 / 4 / 2 / 1, and the
            // small ones re-walk the sources for few cells' worth of stores (14 transforms = 8 + 4 + 2: 0.52 of
            // the HBM roofline against 0.58 for 8 or 16); large grids therefore take groups of whole eights
            static const int quant = std::getenv("FFTVIS_HIP_GROUP_QUANT") ? std::atoi(std::getenv("FFTVIS_HIP_GROUP_QUANT")) : 8;
            const int cq = std::max(1, quant / std::max(tg, 1));  // channels per `quant` transforms
            if (quant > 1 && mb >= 64.0 && b - a > cq && b < f1) b = a + (b - a) / cq * cq;
            g.emplace_back(a, b);
            a = b;
        }
        return g;
    }

    // ---- host output, overlapped (reference cpu_simulate.py:843-854 returns a host array) ---------------------
    // A block of visibilities is 10 GB at C3.  Copied after the last kernel into fresh pageable memory it moves at
    // ~16 GB/s (first touch of every page included: 0.6 s after 1.6 s of compute).  Instead a helper thread pins
    // the caller's array in place (HostPin: parallel first touch, then hipHostRegister) while this thread queues the run,
    // every time step's last kernel records an event, and as soon as the array is pinned
    // the queueing thread issues, on a separate stream, one asynchronous copy per (channel, finished time step)
    // behind that step's event (53 GB/s, PCIe Gen5).  Only the last step's copy (0.5 GB, 10 ms) is left when the
    // kernels end.  FFTVIS_HIP_D2H_OVERLAP=0 or a block under 64 MiB keeps the single copy; so does a buffer the
    // driver refuses to pin.
    static size_t drain_min_bytes() {
        const char *e = std::getenv("FFTVIS_HIP_D2H_OVERLAP");
        if (e && std::atoi(e) == 0) return (size_t)-1;
        const char *m = std::getenv("FFTVIS_HIP_D2H_MIN_BYTES");
        return m ? (size_t)std::atof(m) : ((size_t)64 << 20);
    }
    hipEvent_t drain_event(size_t k) {
        while (drain_events.size() <= k) {
            hipEvent_t e;
            FV_HIP(hipEventCreateWithFlags(&e, std::getenv("FFTVIS_HIP_DEBUG_DRAIN") ? hipEventDefault : hipEventDisableTiming));
            drain_events.push_back(e);
        }
        return drain_events[k];
    }
    struct HostPin {
        std::thread th;
        std::atomic<int> state{0};  // 0 pinning, 1 pinned, -1 refused (locked-memory limit, exotic mapping)
        std::vector<std::pair<char *, size_t>> pieces;  // registered so far (helper thread only, until joined)
        double t_pinned = 0;  // seconds after start() (FFTVIS_HIP_DEBUG_DRAIN)
        hipStream_t *drain_on = nullptr;  // the stream copies into the pinned pieces are queued on (the owner's copy stream)
        std::chrono::steady_clock::time_point t0;
        double since() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
        // Pinning a FRESH array is slow because every page is touched for the first time inside the call, by one thread
        // (22 GB/s; 16.6 GB/s for a plain memset) -- the same pages touched by 16 threads first take 175-230 GB/s, and
        // registering touched memory 480 GB/s (measured, 4 GB).  So: a pool of threads writes one byte into every page
        // of the caller's array (its content is about to be overwritten by the block anyway), then the array is
        // registered in pieces cut at the multiples of 256 MiB of the address space (no page is registered twice; a copy
        // is split at the same addresses, drain_flush; short calls, so that a cold handle's allocations are not held up
        // behind the driver lock).  10 GB: pinned 0.08 s after the start instead of 0.45 s.
        static constexpr uintptr_t PIECE = (uintptr_t)256 << 20;
        static int touch_threads() {
            const unsigned hw = std::thread::hardware_concurrency();
            return (int)std::max(1u, std::min(16u, hw / 4));
        }
        void mark() { t0 = std::chrono::steady_clock::now(); }  // the clock of since() / t_pinned: the run's start
        // The destination is nseg runs of seg_bytes, seg_stride bytes apart (one run: a contiguous array; several: a
        // block inside a larger array, fv_sim_run_into).  shared: other processes write between and around the runs --
        // the first touch then only READS (a write could overwrite what another rank has already delivered; reading
        // faults a shared mapping's pages in just as well) and every run is registered on its own.
        void start(int device, void *ptr, size_t nseg, size_t seg_bytes, size_t seg_stride, bool shared, bool keep_clock = false) {
            if (!keep_clock) mark();
            if (nseg > 1 && seg_stride == seg_bytes) {
                seg_bytes *= nseg;
                nseg = 1;
            }
            th = std::thread([this, device, ptr, nseg, seg_bytes, seg_stride, shared] {
                (void)hipSetDevice(device);
                char *base = static_cast<char *>(ptr);
                {  // first touch, in parallel: thread i takes the i-th share of every run
                    const int nt = touch_threads();
                    std::vector<std::thread> pool;
                    const size_t share = (seg_bytes + nt - 1) / nt;
                    for (int i = 0; i < nt; ++i)
                        pool.emplace_back([=] {
                            for (size_t sg = 0; sg < nseg; ++sg) {
                                char *p = base + sg * seg_stride;
                                volatile char *q = p + std::min(seg_bytes, share * i);
                                volatile char *stop = p + std::min(seg_bytes, share * (i + 1));
                                if (shared) {
                                    char sink = 0;
                                    for (; q < stop; q += 4096) sink ^= *q;
                                    (void)sink;
                                } else {
                                    for (; q < stop; q += 4096) *q = 0;
                                }
                            }
                        });
                    for (std::thread &t : pool) t.join();
                }
                bool ok = true;
                // FFTVIS_HIP_PIN_FAIL_AFTER = n (tests): the (n + 1)-th registration is refused
                const char *ef = std::getenv("FFTVIS_HIP_PIN_FAIL_AFTER");
                const long fail_after = ef ? std::atol(ef) : -1;
                for (size_t sg = 0; ok && sg < nseg; ++sg) {
                char *p = base + sg * seg_stride, *end = p + seg_bytes;
                while (ok && p < end) {
                    const uintptr_t stop = (reinterpret_cast<uintptr_t>(p) / PIECE + 1) * PIECE;
                    char *q = std::min(end, reinterpret_cast<char *>(stop));
                    if ((fail_after < 0 || (long)pieces.size() < fail_after) &&
                        hipHostRegister(p, (size_t)(q - p), hipHostRegisterDefault) == hipSuccess) {
                        pieces.push_back({p, (size_t)(q - p)});
                        p = q;
                    } else {
                        (void)hipGetLastError();
                        ok = false;
                    }
                }
                }
                if (!ok) {
                    // refused part-way (locked-memory limit): the caller's array must not stay HALF pinned -- the fallback
                    // is ONE pageable copy over the whole range, and a range that is part registered, part pageable may be
                    // taken for pinned as a whole.  Nothing has been copied into the pieces yet (copies are only queued
                    // once state is 1), so they are simply released before the refusal is published.
                    for (auto &pc : pieces) (void)hipHostUnregister(pc.first);
                    pieces.clear();
                }
                t_pinned = since();
                state.store(ok ? 1 : -1, std::memory_order_release);
            });
        }
        bool pinned() const { return state.load(std::memory_order_acquire) == 1; }
        bool wait() {
            if (th.joinable()) th.join();
            return pinned();
        }
        ~HostPin() {  // also on the error paths: never leave the caller's memory pinned
            if (th.joinable()) th.join();
            // an exception may unwind past copies that are still in flight into the pieces: they end first
            if (!pieces.empty() && drain_on && *drain_on) (void)hipStreamSynchronize(*drain_on);
            for (auto &pc : pieces) (void)hipHostUnregister(pc.first);
        }
    };
    // queue the copies of the finished time steps items[done ...) behind their events
    void drain_flush(void *out, const cplx<T> *dout, int nt, int nf, int64_t per_tf, const std::vector<DrainItem> &items,
                     size_t &done, int64_t out_fs) {
        if (!copy_stream) FV_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        cplx<T> *hout = static_cast<cplx<T> *>(out);
        for (; done < items.size(); ++done) {
            const DrainItem &it = items[done];
            FV_HIP(hipStreamWaitEvent(copy_stream, it.ev, 0));
            for (int f = 0; f < nf; ++f) {
                const int64_t off = ((int64_t)f * nt + it.t) * per_tf;
                // split where the pinned pieces meet (HostPin): a copy must lie inside one registration
                char *dst = reinterpret_cast<char *>(hout + (int64_t)f * out_fs + (int64_t)it.t * per_tf);
                const char *src = reinterpret_cast<const char *>(dout + off);
                size_t left = sizeof(cplx<T>) * (size_t)it.n * per_tf;
                while (left) {
                    const uintptr_t stop = (reinterpret_cast<uintptr_t>(dst) / HostPin::PIECE + 1) * HostPin::PIECE;
                    const size_t n = std::min<size_t>(left, stop - reinterpret_cast<uintptr_t>(dst));
                    FV_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, copy_stream));
                    dst += n;
                    src += n;
                    left -= n;
                }
            }
        }
    }

    // one asynchronous copy of a whole block into a pinned caller array, split where the pinned pieces meet
    // the same for a block whose channels are out_fs elements apart at the destination: one run per channel
    void copy_block_to_host(void *out, const cplx<T> *dout, int nf, int64_t run_elems, int64_t out_fs, bool pinned, hipStream_t on) {
        if (out_fs == run_elems) {
            if (pinned)
                copy_block_pinned(out, dout, sizeof(cplx<T>) * (size_t)nf * run_elems, on);
            else
                FV_HIP(hipMemcpyAsync(out, dout, sizeof(cplx<T>) * (size_t)nf * run_elems, hipMemcpyDeviceToHost, on));
            return;
        }
        if (!pinned) {
            FV_HIP(hipMemcpy2DAsync(out, sizeof(cplx<T>) * (size_t)out_fs, dout, sizeof(cplx<T>) * (size_t)run_elems,
                                    sizeof(cplx<T>) * (size_t)run_elems, (size_t)nf, hipMemcpyDeviceToHost, on));
            return;
        }
        for (int f = 0; f < nf; ++f)
            copy_block_pinned(static_cast<cplx<T> *>(out) + (int64_t)f * out_fs, dout + (int64_t)f * run_elems,
                              sizeof(cplx<T>) * (size_t)run_elems, on);
    }
    void copy_block_pinned(void *out, const void *dout, size_t bytes, hipStream_t on) {
        char *dst = static_cast<char *>(out);
        const char *src = static_cast<const char *>(dout);
        while (bytes) {
            const uintptr_t stop = (reinterpret_cast<uintptr_t>(dst) / HostPin::PIECE + 1) * HostPin::PIECE;
            const size_t n = std::min<size_t>(bytes, stop - reinterpret_cast<uintptr_t>(dst));
            FV_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, on));
            dst += n;
            src += n;
            bytes -= n;
        }
    }

    void run(int t0, int t1, int f0, int f1, void *out, int out_on_device) override {
        FV_HIP(hipSetDevice(device));
        FV_REQUIRE(nsrc >= 0 && !rots.empty() && !freqs.empty() && nbls > 0 && !pairs.empty(),
                   "engine not fully configured");
        FV_REQUIRE(0 <= t0 && t0 <= t1 && t1 <= (int)rots.size(), "time range");
        FV_REQUIRE(0 <= f0 && f0 <= f1 && f1 <= (int)freqs.size(), "freq range");
        if (mhist_log.size() > 65536) mhist_log.clear();  // nobody asked for the statistics of those runs
        FV_REQUIRE((int)freqs.size() == nfreq_cat, "flux frequency axis != freqs");
        for (const Beam &b : beams) FV_REQUIRE(b.kind >= 0, "beam not set");
        if (type1) {
            run_type1(t0, t1, f0, f1, out, out_on_device);
            return;
        }
        const int nt = t1 - t0, nf = f1 - f0;
        // destination layout of this run (fv_sim_run_into), consumed here
        const int64_t out_fs = out_f_stride ? out_f_stride : (int64_t)nt * tpol * nbls;
        const bool shared = out_shared != 0;
        out_f_stride = 0;
        out_shared = 0;
        FV_REQUIRE(out_fs >= (int64_t)nt * tpol * nbls, "fv_sim_run_into: the channel stride is shorter than a channel's run");
        // height terms instead of a third grid dimension (see wt_K)
        double xc[3], X[3];
        source_box(xc, X);
        wt_K = 0;
        wt_a = 0.0;
        if (!coplanar && !std::getenv("FFTVIS_HIP_NO_WTERM")) {
            double fmax = 0, bz = 0;
            for (int f = f0; f < f1; ++f) fmax = std::max(fmax, std::fabs(freqs[f]));
            for (const Pair &p : pairs)
                if (p.n) bz = std::max(bz, p.Bs[2]);
            const double a = X[2] * fmax * bz;  // largest |(z - zc) s_z|
            int K = 1;
            double term = a;  // 2 (a / 2)^K / K!: bound of the first neglected coefficient 2 |J_K|
            while (term > 0.1 * eps && K < 64) {
                ++K;
                term *= 0.5 * a / K;
            }
            const char *ek = std::getenv("FFTVIS_HIP_WTERM_MAX");
            if (K <= (ek ? std::atoi(ek) : 16)) {
                wt_K = K;
                wt_zc = xc[2];
                wt_zh = X[2];
                wt_a = a;
            }
        }
        wt_k0 = 0;
        wt_k1 = wt_K;
        wt_eps_l[0] = wt_eps_l[1] = 0.0;
        if (wt_K >= 3 && sizeof(T) == 8 && this->sigma == 2.0 && !std::getenv("FFTVIS_HIP_NO_WTERM_LIGHT")) {
            std::vector<double> ck(wt_K);
            double c = 1.0;  // (a / 2)^k / k!
            for (int k = 0; k < wt_K; ++k) {
                ck[k] = (k ? 2.0 : 1.0) * c;
                c *= 0.5 * wt_a / (k + 1);
            }
            for (int k0 = 1; k0 + 2 <= wt_K; ++k0) {  // at least two light terms
                double sl = 0;
                for (int k = k0; k < wt_K; ++k) sl += ck[k];
                const double el = 0.3 * eps / sl;
                if (el >= 1e-7) {
                    wt_k0 = k0;
                    wt_eps_l[0] = std::min(el, 1e-2);
                    break;
                }
            }
            // a second class for the tail, where it can run at 1e-4 or looser: half of the light budget each
            if (wt_k0 > 0 && !std::getenv("FFTVIS_HIP_WTERM_ONE_LIGHT_CLASS")) {
                for (int k1 = wt_k0 + 1; k1 + 2 <= wt_K; ++k1) {
                    double s2 = 0, s1 = 0;
                    for (int k = k1; k < wt_K; ++k) s2 += ck[k];
                    for (int k = wt_k0; k < k1; ++k) s1 += ck[k];
                    const double e2 = 0.15 * eps / s2, e1 = 0.15 * eps / s1;
                    if (e2 >= 1e-4 && e1 >= 1e-7) {
                        wt_k1 = k1;
                        wt_eps_l[0] = std::min(e1, 1e-2);
                        wt_eps_l[1] = std::min(e2, 1e-2);
                        break;
                    }
                }
            }
        }
        const int D = wt_K ? 2 : dim();
        run_D = D;
        st[15] = wt_K;
        const int64_t per_tf = (int64_t)tpol * nbls;   // elements per (freq, time)
        const size_t out_bytes = sizeof(cplx<T>) * (size_t)nf * nt * per_tf;
        cplx<T> *dout;
        if (out_on_device) {
            dout = (cplx<T> *)out;
        } else {
            d_out.reserve(std::max<size_t>(out_bytes, 16));
            dout = d_out.as<cplx<T>>();
        }
        // Baselines not covered by any pair stay zero (reference zero-initialises, :909-911).
        FV_HIP(hipMemsetAsync(dout, 0, out_bytes, stream));
        reserve_mhist(sizeof(int) * rots.size() * std::max(1, src_chunks));
        const size_t run_bytes = sizeof(cplx<T>) * (size_t)nt * per_tf;
        // (a block inside a larger array is pinned run by run: only worth it -- and only safe against two runs meeting in
        // one page -- when the runs are long)
        const bool pinnable = out_fs == (int64_t)nt * per_tf || (run_bytes >= ((size_t)1 << 20) && (size_t)(out_fs - (int64_t)nt * per_tf) * sizeof(cplx<T>) >= 8192);
        const bool drain = !out_on_device && out_bytes >= drain_min_bytes() && pinnable;
        std::vector<DrainItem> drain_items;
        size_t drained = 0;
        // the helper touches and pins the caller's array once the first unit is queued: started at once, its sixteen page-
        // faulting threads slowed the main thread's set-up and first launches (first unit queued after 95 ms instead of 50)
        HostPin pin;
        pin.drain_on = &copy_stream;
        pin.mark();
        bool pin_started = false;
        static const bool pin_early = std::getenv("FFTVIS_HIP_PIN_EARLY") != nullptr;
        if (drain && pin_early) {
            pin.start(device, out, (size_t)nf, run_bytes, sizeof(cplx<T>) * (size_t)out_fs, shared, true);
            pin_started = true;
        }

        int64_t pol_off[16] = {0};
        if (polarized)
            for (int r = 0; r < 4; ++r) pol_off[r] = (int64_t)((r % 2) * 2 + r / 2) * nbls;

        // source chunks: chunk c covers catalog sources [c csz, min(nsrc, (c + 1) csz)); the compacted
        // per-time arrays hold source_buffer x csz sources
        const int nch = (int)std::max<int64_t>(1, std::min<int64_t>(src_chunks, nsrc));
        const int64_t csz = std::max<int64_t>(cdiv(nsrc, nch), 1);
        const int64_t cap = std::max<int64_t>((int64_t)std::ceil(csz * source_buffer), 1);
        const int nblk = (int)cdiv(csz, 256);

        // Upsampling factor "auto" (fv_sim_create upsampfac = 0): sigma = 1.25 shrinks the fine grid and
        // all FFT work by (2 / 1.25)^D at the price of a kernel 13-14 cells wide instead of 9 (every
        // source and target costs ~2x in 2-D), with NUFFT errors at or below sigma = 2's down to
        // eps ~ 1e-8 (fv_eskernel.h).  It pays when the FFT dominates: C3 (8192^2 cells, 1.1e5 points
        // per transform) 3.06 -> 1.57 s per step; C2 (1024 x 512, 5.7e3) would lose, 1.38 -> 1.58 ms.
        // Hermitian packing (k_interp<.., HERM>): a pair whose two beams are the same has Hermitian
        // strengths -- c_00, c_11 real, c_10 = conj(c_01) -- so two transforms per frequency (c_00 + i c_11,
        // c_01) evaluated at the baseline and at its mirror image give all four products: half the
        // spread, FFT and grid traffic of a polarized run.  The mirror targets need a box that is
        // symmetric about 0; taken when that box costs at most 1.5x the cells of the tight one and
        // the grid is large (small grids keep four transforms and the fused gather).  In eigenbeam mode the
        // diagonal (k, k) terms qualify.  FFTVIS_HIP_NO_HERMITIAN=1 turns it off.
        {
            const bool herm_off = std::getenv("FFTVIS_HIP_NO_HERMITIAN") != nullptr;  // read per run: tests flip it
            const KerParams k2 = make_kernel(eps, 2.0);
            double fmax = 0;
            for (int f = f0; f < f1; ++f) fmax = std::max(fmax, std::fabs(freqs[f]));
            for (Pair &p : pairs) {
                p.herm = 0;
                if (!polarized || herm_off || p.n == 0) continue;
                // 1: same beam on both sides (Hermitian strengths; eigenbeams: the (k, k) terms);
                // 2: two different beams whose Jones matrices are real, unpolarized sky (all products real)
                const int mode = p.bi == p.bj ? 1 : (!pol_sky && beams[p.bi].real_valued && beams[p.bj].real_valued ? 2 : 0);
                if (!mode) continue;
                double cs = 1.0, ct = 1.0;
                for (int d = 0; d < D; ++d) {
                    DimGeom gs, gt;
                    gs.X = gt.X = X[d];
                    gs.B = p.Bs[d];
                    gt.B = p.B[d];
                    set_dim_geom(gs, 2.0, k2.w, fmax, d == D - 1);
                    set_dim_geom(gt, 2.0, k2.w, fmax, d == D - 1);
                    cs *= gs.n2;
                    ct *= gt.n2;
                }
                p.herm = cs >= 4.0e6 && cs <= 1.5 * ct ? mode : 0;
            }
            // reference_compat off, eigenbeams: an off-diagonal pair that is not packed gathers its (l, k) term
            // at -b: its targets need the symmetric box too
            for (Pair &p : pairs) p.mirror = nbasis && !reference_compat && p.bi != p.bj && !p.herm && p.n > 0;
            // redundant baselines -> one gather target each; the tolerance follows the engine's eps and the highest
            // frequency it knows (not the block's: blocks of a sharded run then agree on the runs)
            double fall = 0;
            for (double f : freqs) fall = std::max(fall, std::fabs(f));
            const double tol = 1e-3 * eps / (2.0 * M_PI * std::max(fall, 1.0));
            for (Pair &p : pairs) {
                build_unique(p, tol);
                pair_mirror_runs(p, tol);
            }
        }
        int tg_max = 1;  // transforms per frequency on the grid, largest over the pairs
        for (const Pair &p : pairs)
            if (p.n) tg_max = std::max(tg_max, p.herm ? 2 : tpol);
        double sigma = this->sigma;
        if (sigma == 0.0) {
            const KerParams k2 = make_kernel(eps, 2.0);
            double fmax = 0, cells2 = 1.0;
            for (int f = f0; f < f1; ++f) fmax = std::max(fmax, std::fabs(freqs[f]));
            int64_t nmax = 0;
            for (const Pair &p : pairs) nmax = std::max<int64_t>(nmax, p.n);
            for (int d = 0; d < D; ++d) {
                DimGeom g;
                g.X = X[d];
                double Bm = 0;
                for (const Pair &p : pairs) Bm = std::max(Bm, p.box_B()[d]);
                g.B = Bm;
                set_dim_geom(g, 2.0, k2.w, fmax, d == D - 1);
                cells2 *= g.n2;
            }
            const double points = 0.5 * (double)nsrc + (double)nmax;
            // accuracy floor of sigma = 1.25: the kernel transform falls by ~e^{-w/2} per dimension across
            // the band and rounding is amplified by that factor at band-edge targets -- ~1e-8 in fp64; in
            // fp32 it matches sigma = 2 down to eps = 1e-4 (HERA-350, top of the band: worst baseline
            // 5.8e-4 vs 9.9e-4, rel. l2 4.1e-5 vs 6.4e-5) and falls behind at 1e-5
            // (3-D: one more dimension of amplification -- 4e-8 seen at eps 2.5e-9 -- so ten times higher)
            const double eps_floor = (sizeof(T) == 8 ? 1e-8 : 1e-4) * (D == 3 ? 10.0 : 1.0);
            // measured (2-D): 8192^2 grids win with 1.25 from 1e5 sources (3.06 -> 1.57 s) up to 4e6 per
            // time step (31.1 -> 29.7 ms per 16-channel slice, ~30 cells per point); a 1024 x 512 grid
            // loses slightly even with 1e3 sources (its kernels are latency-bound, a smaller grid buys
            // little): so large grids only, and not when points outnumber the cells they save
            const double per_point = D == 2 ? 30.0 : 200.0;
            sigma = eps >= eps_floor && cells2 >= 4.0e6 && cells2 >= per_point * points ? 1.25 : 2.0;
        }
        sigma_run = sigma;
        st[10] = sigma;
        // grid-buffer cells per transform at the top frequency, for the grouping heuristic
        double cells_top = 1.0;
        {
            KerParams k = make_kernel(eps, sigma);
            double fmax = 0;
            for (int f = f0; f < f1; ++f) fmax = std::max(fmax, std::fabs(freqs[f]));
            double na[3] = {1, 1, 1}, no[3] = {1, 1, 1};
            for (int d = 0; d < D; ++d) {
                DimGeom g;
                g.X = X[d];
                double Bm = 0;
                for (const Pair &p : pairs) Bm = std::max(Bm, p.box_B()[d]);
                g.B = Bm;
                set_dim_geom(g, sigma, k.w, fmax, d == D - 1);
                na[d] = g.na;
                no[d] = g.no;
            }
            cells_top = 2.0 * std::max({na[2] * na[1] * na[0], na[2] * na[1] * no[0], na[2] * no[0] * no[1],
                                         no[2] * no[0] * no[1]});
        }
        const auto groups = freq_groups(f0, f1, cells_top, tg_max);

        // two lanes while a group's grid buffers are small (launch-bound regime), else one
        int max_ntrans = 1;
        for (const auto &grp : groups) max_ntrans = std::max(max_ntrans, (grp.second - grp.first) * tg_max);
        // Two lanes always (two sets of per-time scratch and grid buffers; consecutive time steps alternate), memory
        // permitting.  Small grids (launch-bound) run them pipelined, see below.  Large grids (C3: 6 GiB of grid per
        // launch) run them FREELY on two streams of equal priority: the kernels of two time steps then share the
        // dispatcher like the kernels of two processes do -- a row pass of one step beside the spread or the gather of
        // the other, compute-bound waves beside memory-bound ones -- which is what two ranks on one GPU had over one
        // (843 against 883 ms per C3 step): 883 -> 844 ms in-process.  (With the second stream at a lower priority it
        // only ever filled the first one's tails: 868.)  Kernel durations measured in this mode are those of kernels
        // sharing the GPU.  FFTVIS_HIP_LANES=1: one stream.
        const bool big_grids = cells_top * sizeof(cplx<T>) * max_ntrans > 1.5 * 1024 * 1024 * 1024;
        const char *el = std::getenv("FFTVIS_HIP_LANES");
        int nlanes = el ? std::atoi(el) : 2;
        if (!el && big_grids) {  // a second set of grid buffers must fit comfortably
            size_t mfree = 0, mtotal = 0;
            FV_HIP(hipMemGetInfo(&mfree, &mtotal));
            if (4.0 * cells_top * sizeof(cplx<T>) * max_ntrans > 0.5 * (double)mtotal) nlanes = 1;
        }
        const char *ep = std::getenv("FFTVIS_HIP_PIPE");
        const bool pipe_wanted = ep ? std::atoi(ep) != 0 : !big_grids;
        nlanes = std::max(1, std::min(pipe_wanted ? 2 : 4, std::min(nlanes, nt)));
        if (timing_level == 2) nlanes = 1;  // per-family event brackets only make sense on one stream
        // Two lanes, pipelined (default): every big kernel runs on the main (high-priority) stream,
        // one time step after the other, so kernel durations stay uncontended; the dozen tiny
        // latency-bound preparation kernels of step t+1 (rotation, horizon cut, bin sort, weight
        // tables) run on a low-priority stream beside step t's big kernels and fill their ramps
        // and tails.  FFTVIS_HIP_PIPE=0: the two lanes run freely on two streams instead.
        const bool pipe = nlanes > 1 && pipe_wanted;
        // Gang mode (pipelined 2-D runs): two consecutive time steps share one launch each of the
        // spread and of every FFT pass (grid.y = 2: same geometry, their own sources and grids), which
        // halves the kernel boundaries per time step and doubles the workgroups that hide each other's
        // latency chains and tails.  Two pairs of lanes alternate, so that the preparation of the next
        // pair still runs beside this pair's big kernels.  FFTVIS_HIP_GANG=0 turns it off.
        const char *eg = std::getenv("FFTVIS_HIP_GANG");
        const bool gang = pipe && D == 2 && nt >= 2 && timing_level != 2 && !(eg && std::atoi(eg) == 0);
        const int nlanes_used = gang ? 4 : nlanes;
        // Lane scratch outlives a run: with a device-side output buffer nothing synchronises between
        // two fv_sim_run calls, so the "last big kernels of this lane" events carry over (the next
        // run's first preparation waits for them) and the lane rotation continues where the previous
        // run stopped -- its first unit then takes the lanes that have been idle longest and prepares
        // beside the previous run's last big kernels.  A change of mode drains the streams instead.
        const int mode = gang ? 2 : pipe ? 1 : 0;
        st[16] = nlanes;
        st[17] = mode;
        if (mode != lane_mode) {
            FV_HIP(hipStreamSynchronize(stream));
            FV_HIP(hipStreamSynchronize(prep_stream));
            for (int li = 1; li < 4; ++li)
                if (lanes[li].stream && lanes[li].own_stream) FV_HIP(hipStreamSynchronize(lanes[li].stream));
            for (Lane &L : lanes) L.heavy_pending = false;
            lane_mode = mode;
            lane_serial = 0;
        }
        if (nlanes > 2 && !pipe) {  // FFTVIS_HIP_LANES = 3 | 4: their streams, at the main stream's priority
            int prio_least = 0, prio_greatest = 0;
            FV_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
            for (int li = 2; li < nlanes; ++li)
                if (!lanes[li].stream || !lanes[li].own_stream) {
                    FV_HIP(hipStreamCreateWithPriority(&lanes[li].stream, hipStreamNonBlocking, prio_greatest));
                    lanes[li].own_stream = true;
                }
        }
        {  // (sigma = 1.25 pays on large grids only, as in the automatic choice; FFTVIS_HIP_WTERM_LIGHT_CELLS moves the bound: tests)
            const char *elc = std::getenv("FFTVIS_HIP_WTERM_LIGHT_CELLS");
            if (wt_k0 > 0 && cells_top < (elc ? std::atof(elc) : 4.0e6)) wt_k0 = 0;
        }
        if (wt_k0 == 0) wt_k1 = wt_K;
        st[18] = wt_k0;
        st[19] = wt_k0 > 0 && wt_k1 < wt_K ? wt_k1 : 0;
        const int nlc = wt_k0 == 0 ? 0 : wt_k1 < wt_K ? 2 : 1;  // light classes of this run
        for (int li = 0; li < nlanes_used; ++li) {
            Lane &L = lanes[li];
            // height terms: term k enters with weight |c_k| <= 2 (a / 2)^k / k! (sum <= 2 e^{a/2} - 1), each with the
            // transform's relative error -- the run's own plan takes eps / (2 e^{a/2} - 1) so that the sum keeps eps
            const double eps_plan = wt_K ? std::max(eps / (2.0 * std::exp(0.5 * wt_a) - 1.0), sizeof(T) == 8 ? 1e-14 : 1e-7) : eps;
            if (!L.nufft || L.nufft->dim != D || L.nufft->sigma != sigma || L.nufft->eps != eps_plan)
                L.nufft.reset(new Nufft3<T>(D, eps_plan, sigma, li < 2 || !pipe ? L.stream : stream));
            L.nufft->err_oob = d_err.as<int>();
            // the sources are 2 pi x (projections of unit vectors onto the array plane): inside a disc whatever the box
            L.nufft->disc_radius = (D == 2 || L.nufft->zdirect) && !std::getenv("FFTVIS_HIP_NO_DISC") ? 2.0 * M_PI : 0.0;
            L.nufft->transpose_flipped = !reference_compat;
            if (li > 0) L.nufft->order_cache = lanes[0].nufft->order_cache;  // one table per grid size for all lanes
            for (int c = 0; c < nlc; ++c) {  // the light height terms' plans
                if (!L.nufft_l[c] || L.nufft_l[c]->eps != wt_eps_l[c])
                    L.nufft_l[c].reset(new Nufft3<T>(2, wt_eps_l[c], 1.25, li < 2 || !pipe ? L.stream : stream));
                L.nufft_l[c]->err_oob = d_err.as<int>();
                L.nufft_l[c]->disc_radius = L.nufft->disc_radius;
                L.nufft_l[c]->transpose_flipped = !reference_compat;
                if (li > 0) L.nufft_l[c]->order_cache = lanes[0].nufft_l[c]->order_cache;
                L.binned_ti_l[c] = -1;
            }
            L.d_xyz.reserve(sizeof(T) * 3 * cap);
            L.d_az.reserve(sizeof(T) * cap);
            L.d_za.reserve(sizeof(T) * cap);
            L.d_srcidx.reserve(sizeof(int) * cap);
            L.d_blockcnt.reserve(sizeof(int) * (nblk + 1));
            L.d_blockoff.reserve(sizeof(int) * (nblk + 1));
            L.binned_ti = -1;
        }
        const bool dbg_t = std::getenv("FFTVIS_HIP_DEBUG_DRAIN") != nullptr;
        if (dbg_t) std::fprintf(stderr, "run: set-up done %.3f s (lanes, unique targets, groups)\n", pin.since());
        // Size every lane's grid and strength buffers for the largest (frequency group, beam pair) of this run now,
        // before anything is queued (Nufft3::plan_buffer_cells): no reallocation -- a device synchronisation each --
        // while the first time step runs.
        {
            int64_t need = 0, need_str = 0;
            int na_max[3] = {8, 8, 8}, n2_max[3] = {64, 64, 64};
            for (const auto &grp : groups) {
                double smax = 0;
                for (int f = grp.first; f < grp.second; ++f) smax = std::max(smax, std::fabs(freqs[f]));
                for (const Pair &pr : pairs) {
                    if (pr.n == 0) continue;
                    const int ntrans = (grp.second - grp.first) * (pr.herm ? 2 : tpol);
                    for (double sl : {0.0, -1.0}) {  // with and without a column plan (its geometry takes no grid slack)
                        lanes[0].nufft->grid_slack = sl;
                        need = std::max(need, lanes[0].nufft->plan_buffer_cells(X, pr.box_B(), smax, na_max, n2_max) * ntrans);
                    }
                    need_str = std::max<int64_t>(need_str, ntrans);
                }
            }
            for (int li = 0; li < nlanes_used; ++li) {
                lanes[li].nufft->reserve_buffers(need, na_max, n2_max);
                lanes[li].nufft->strengths_buffer_reserve(cap, (int)need_str);
            }
            for (int c = 0; c < nlc; ++c) {  // the light height terms' plans: their own (smaller) grids
                int64_t need_l = 0;
                int na_l[3] = {8, 8, 8}, n2_l[3] = {64, 64, 64};
                for (const auto &grp : groups) {
                    double smax = 0;
                    for (int f = grp.first; f < grp.second; ++f) smax = std::max(smax, std::fabs(freqs[f]));
                    for (const Pair &pr : pairs) {
                        if (pr.n == 0) continue;
                        const int ntrans = (grp.second - grp.first) * (pr.herm ? 2 : tpol);
                        for (double sl : {0.0, -1.0}) {
                            lanes[0].nufft_l[c]->grid_slack = sl;
                            need_l = std::max(need_l, lanes[0].nufft_l[c]->plan_buffer_cells(X, pr.box_B(), smax, na_l, n2_l) * ntrans);
                        }
                    }
                }
                for (int li = 0; li < nlanes_used; ++li) {
                    lanes[li].nufft_l[c]->reserve_buffers(need_l, na_l, n2_l);
                    lanes[li].nufft_l[c]->strengths_buffer_reserve(cap, (int)need_str);
                }
            }
            // column plans of every (group, pair), from the geometry the run will set (large 2-D grids only)
            col_plan_of.assign(groups.size() * pairs.size(), nullptr);
            // plans of earlier target sets / groupings pile up in a long-lived handle: start over now and then (here,
            // before this run takes pointers into the list, and after everything an earlier run queued has finished)
            if (col_plans.size() > 512) {
                FV_HIP(hipDeviceSynchronize());
                col_plans.clear();
            }
            if ((D == 2 || lanes[0].nufft->zdirect) && !std::getenv("FFTVIS_HIP_NO_COLUMN_PLAN")) {
                Nufft3<T> *n0 = lanes[0].nufft.get();
                for (size_t gi = 0; gi < groups.size(); ++gi) {
                    double smax = 0;
                    for (int f = groups[gi].first; f < groups[gi].second; ++f) smax = std::max(smax, std::fabs(freqs[f]));
                    for (size_t pi = 0; pi < pairs.size(); ++pi) {
                        const Pair &pr = pairs[pi];
                        if (pr.n == 0) continue;
                        n0->grid_slack = 0.0;  // a plan's geometry: no grid slack (set_dim_geom)
                        n0->set_geometry(xc, X, pr.box_c(), pr.box_B(), smax);
                        if (!n0->columns_possible() || n0->geo.cells_o() < 4000000) continue;
                        col_plan_of[gi * pairs.size() + pi] = column_plan((int)pi, pr, groups[gi].first, groups[gi].second, n0);
                    }
                }
                FV_HIP(hipStreamSynchronize(n0->stream));  // the table kernels of these set_geometry calls are done before the run's own
            }
            for (int c = 0; c < 2; ++c) col_plan_of_l[c].assign(groups.size() * pairs.size(), nullptr);
            for (int c = 0; c < nlc && !std::getenv("FFTVIS_HIP_NO_COLUMN_PLAN"); ++c) {
                Nufft3<T> *n0 = lanes[0].nufft_l[c].get();
                for (size_t gi = 0; gi < groups.size(); ++gi) {
                    double smax = 0;
                    for (int f = groups[gi].first; f < groups[gi].second; ++f) smax = std::max(smax, std::fabs(freqs[f]));
                    for (size_t pi = 0; pi < pairs.size(); ++pi) {
                        const Pair &pr = pairs[pi];
                        if (pr.n == 0) continue;
                        n0->grid_slack = 0.0;
                        n0->set_geometry(xc, X, pr.box_c(), pr.box_B(), smax);
                        if (!n0->columns_possible() || n0->geo.cells_o() < 4000000) continue;
                        col_plan_of_l[c][gi * pairs.size() + pi] = column_plan((int)pi, pr, groups[gi].first, groups[gi].second, n0);
                    }
                }
                FV_HIP(hipStreamSynchronize(n0->stream));
            }
        }
        if (dbg_t) std::fprintf(stderr, "run: buffers and column plans %.3f s\n", pin.since());
        if (nlanes > 1 && !pipe) {  // the other lanes start after the output memset queued on the main stream
            FV_HIP(hipEventRecord(ev_start, stream));
            for (int li = 1; li < nlanes; ++li) FV_HIP(hipStreamWaitEvent(lanes[li].stream, ev_start, 0));
        }

        const int sample_step = std::min(TIMING_STRIDE / 2, nt - 1);  // level-1 timing: this step of every 16
        const Pair *last_pair = nullptr;
        for (const Pair &pr : pairs)
            if (pr.n) last_pair = &pr;
        // only when every unit runs one geometry (one frequency group, one beam pair): with several, the
        // next preparation's set_geometry may rewrite the twiddle tables the FFT passes still read
        int active_pairs = 0;
        for (const Pair &pr : pairs) active_pairs += pr.n > 0;
        const char *erh = std::getenv("FFTVIS_HIP_RIDE_EVENT");
        const bool ride_heavy_done = !(erh && std::atoi(erh) == 0) && groups.size() == 1 && active_pairs == 1;
        for (int tnext = t0, ch = 0; tnext < t1;) {
            // units: (one or two time steps) x source chunk, chunks innermost (cpu_simulate.py:936-939)
            const int nm = gang && tnext + 1 < t1 ? 2 : 1;  // time steps in this unit
            const int tu = tnext;
            const int64_t s0 = (int64_t)ch * csz, sn = std::min<int64_t>(csz, nsrc - s0);
            const bool accumulate = ch > 0;  // later chunks add to the first one's visibilities (:1024,1069)
            const int chunk = ch;
            if (++ch == nch) {
                ch = 0;
                tnext += nm;
            }
            // host output: once the last chunk of these time steps is queued, an event marks them finished
            hipStream_t unit_stream = stream;  // where this unit's big kernels run (free-running lanes: the lane's own)
            auto close_time = [&]() {
                if (!drain || chunk != nch - 1) return;
                const hipEvent_t ev = drain_event(drain_items.size());
                // pipelined and single-lane runs: every big kernel is on `stream`; free-running lanes: all chunks of a time
                // step ran, in order, on its lane's stream -- the copy stream waits for THAT (the lanes never wait for
                // each other)
                FV_HIP(hipEventRecord(ev, unit_stream));
                drain_items.push_back({ev, tu - t0, nm});
                if (pin.pinned()) drain_flush(out, dout, nt, nf, per_tf, drain_items, drained, out_fs);
            };
            if (nsrc == 0 || sn <= 0) {  // nothing above the horizon: the block stays zero (:945-946)
                close_time();
                continue;
            }
            const int64_t unit = lane_serial++;
            Lane *Ls[2];
            if (gang) {
                Ls[0] = &lanes[(unit % 2) * 2];
                Ls[1] = &lanes[(unit % 2) * 2 + 1];
            } else if (pipe) {
                Ls[0] = Ls[1] = &lanes[unit % nlanes];
            } else {
                // free-running lanes: the source chunks of one time step ADD to one another's visibilities, so they stay
                // on one stream, in order
                Ls[0] = Ls[1] = &lanes[tu % nlanes];
                unit_stream = Ls[0]->stream;
            }
            Lane &L0 = *Ls[0];
            const hipStream_t ls = pipe ? stream : L0.stream;        // big kernels
            const hipStream_t ps = pipe ? prep_stream : L0.stream;   // per-time preparation
            bool sampled = false, heavy_recorded = false;
            for (int m = 0; m < nm; ++m) sampled = sampled || (tu + m - t0) % TIMING_STRIDE == sample_step;
            // ---- per-time: rotate, horizon cut, az/za, 2 pi R topo --------------------------
            if (pipe && L0.heavy_pending) FV_HIP(hipStreamWaitEvent(ps, L0.heavy_done, 0));  // lane scratch is free
            const Pair *first_pair = nullptr;
            bool strengths_ahead = false;
            const int64_t M = cap;  // capacity: array stride and launch bound
            const int *Mps[2] = {nullptr, nullptr};
            size_t hist_slot[2] = {0, 0};
            size_t e0 = ev_begin(TM_PREP, ps);
            for (int m = 0; m < nm; ++m) {
                RoctxRange rr("prep");
                Lane &L = *Ls[m];
                Nufft3<T> *nufft = L.nufft.get();
                nufft->stream = ps;
                Mps[m] = horizon_step(L, tu + m, cap, nblk, ps, s0, sn, (int64_t)(tu + m) * nch + chunk);
                if (pipe) {  // the first (group, pair)'s bin sort belongs to the preparation as well
                    for (const Pair &pr : pairs) {
                        if (pr.n == 0 || groups.empty()) continue;
                        double smax0 = 0;
                        for (int f = groups[0].first; f < groups[0].second; ++f)
                            smax0 = std::max(smax0, std::fabs(freqs[f]));
                        {
                            const ColPlan *cpq = col_plan_of[(size_t)(&pr - pairs.data())];  // group 0
                            nufft->grid_slack = cpq && cpq->use ? 0.0 : -1.0;
                        }
                        nufft->set_geometry(xc, X, pr.box_c(), pr.box_B(), smax0);
                        nufft->set_sources(M, L.d_xyz.template as<T>(), L.d_xyz.template as<T>() + cap,
                                           D > 2 ? L.d_xyz.template as<T>() + 2 * cap : nullptr, Mps[m]);
                        L.binned_ti = (tu + m) * nch + chunk;
                        L.binned_serial = nufft->geom_serial;
                        // ... and so do its strengths (beam x coherency, pre-phase): they depend on this
                        // step's sources only, not on the previous step's big kernels
                        launch_strengths(L, pr, groups[0].first, groups[0].second - groups[0].first, M, Mps[m], ps);
                        first_pair = &pr;
                        strengths_ahead = true;
                        break;
                    }
                }
                hist_slot[m] = mhist_log.size();
                mhist_log.push_back({(tu + m) * nch + chunk, 0.0});
            }
            ev_end(e0, ps);
            if (pipe) {
                FV_HIP(hipEventRecord(L0.prep_done, ps));
                FV_HIP(hipStreamWaitEvent(ls, L0.prep_done, 0));
            }
            for (int m = 0; m < nm; ++m) {
                Ls[m]->nufft->stream = ls;
                for (int c = 0; c < nlc; ++c) Ls[m]->nufft_l[c]->stream = ls;
            }

            for (const auto &grp : groups) {
                const int fa = grp.first, fb = grp.second, nfg = fb - fa;
                double smax = 0;
                for (int f = fa; f < fb; ++f) smax = std::max(smax, std::fabs(freqs[f]));
                for (const Pair &pr : pairs) {
                    if (pr.n == 0) continue;
                    const int tg = pr.herm ? 2 : tpol;  // transforms per frequency on the grid
                    const int ntrans = nfg * tg;
                    // height terms (wt_K): one round of strengths -> spread -> FFT -> gather per term, the gather adding
                    // term k with every baseline's own factor; otherwise a single round
                    for (int kt = 0; kt < std::max(1, wt_K); ++kt) {
                    // the plan of this term: the run's own, or the light terms' (wt_k0)
                    const bool light = wt_k0 > 0 && kt >= wt_k0;
                    const int lc = light && kt >= wt_k1 ? 1 : 0;  // its class
                    auto plan_of = [&](Lane &L) { return light ? L.nufft_l[lc].get() : L.nufft.get(); };
                    const std::vector<ColPlan *> &cplans = light ? col_plan_of_l[lc] : col_plan_of;
                    Nufft3<T> *nufft = plan_of(L0);
                    Nufft3<T> *mate = nm == 2 ? plan_of(*Ls[1]) : nullptr;
                    for (int m = 0; m < nm; ++m) {
                        Lane &L = *Ls[m];
                        Nufft3<T> *nf_ = plan_of(L);
                        int &b_ti = light ? L.binned_ti_l[lc] : L.binned_ti;
                        int64_t &b_serial = light ? L.binned_serial_l[lc] : L.binned_serial;
                        // ---- geometry + bin sort (skipped when unchanged since last set) -------
                        RoctxRange rr("prep");
                        size_t e1 = ev_begin(TM_PREP, ls);
                        {
                            const ColPlan *cpq = cplans[(size_t)(&grp - groups.data()) * pairs.size() + (size_t)(&pr - pairs.data())];
                            nf_->grid_slack = cpq && cpq->use ? 0.0 : -1.0;
                        }
                        nf_->set_geometry(xc, X, pr.box_c(), pr.box_B(), smax);
                        if (b_ti != (tu + m) * nch + chunk || b_serial != nf_->geom_serial || nf_->M != M) {
                            nf_->set_sources(M, L.d_xyz.template as<T>(), L.d_xyz.template as<T>() + cap,
                                             D > 2 ? L.d_xyz.template as<T>() + 2 * cap : nullptr, Mps[m]);
                            b_ti = (tu + m) * nch + chunk;
                            b_serial = nf_->geom_serial;
                        }
                        ev_end(e1, ls);
                        // ---- strengths (already queued with the preparation for the first pair) -------
                        if (!(strengths_ahead && &grp == &groups.front() && &pr == first_pair && kt == 0))
                            launch_strengths(L, pr, fa, nfg, M, Mps[m], ls, kt, nf_);
                    }
                    // ---- NUFFT ----------------------------------------------------------
                    {
                        ColPlan *cp = cplans[(size_t)(&grp - groups.data()) * pairs.size() + (size_t)(&pr - pairs.data())];
                        const bool on = cp && cp->use;
                        for (int m = 0; m < nm; ++m) {
                            plan_of(*Ls[m])->arm_columns(on ? cp->tab.template as<int>() : nullptr, on ? cp->xtab.template as<int>() : nullptr, tg,
                                                      on ? cp->ncc : 0, d_err.as<int>() + 3,
                                                      on && cp->omask.p ? cp->omask.template as<unsigned long long>() : nullptr,
                                                      on ? cp->nblk : 0);
                            plan_of(*Ls[m])->col_out_cells = on && cp->omask.p ? cp->out_cells : 0.0;
                        }
                    }
                    {
                    RoctxRange rr("spread");
                    if (timing_level >= 2 || (timing_level == 1 && sampled)) {  // 2 and 3: every launch
                        const size_t e3 = ev_slot(TM_SPREAD);
                        nufft->spread(ntrans, ev_pool[e3].a, ev_pool[e3].b, mate);
                        spread_timed += nm;  // a gang launch serves nm time steps: counted per time step
                    } else if (ride_heavy_done && pipe && &grp == &groups.back() && &pr == last_pair && kt + 1 >= std::max(1, wt_K)) {
                        // the unit's last spread is the last reader of the lanes' per-time arrays (the FFT
                        // passes and the gather work on the grids): its dispatch carries the "lane scratch
                        // is free" event, which saves the main stream a marker packet per unit
                        nufft->spread(ntrans, nullptr, L0.heavy_done, mate);
                        heavy_recorded = true;
                    } else {
                        nufft->spread(ntrans, nullptr, nullptr, mate);
                    }
                    }
                    st[0] += nm;  // launches are counted per (time, frequency group, beam pair)
                    st[1] += (double)nufft->spread_cells() * ntrans * nm;  // cells written (2-D: the blocks inside the source disc)
                    for (int m = 0; m < nm; ++m) mhist_log[hist_slot[m]].second += ntrans;
                    cplx<T> *obase = dout + ((int64_t)(fa - f0) * nt + (tu - t0)) * per_tf;
                    // small 2-D grids: the last FFT pass serves the targets from its LDS tiles (no C
                    // buffer, no gather kernel); the output block was zeroed at the start of the run
                    const bool fused =
                        !nbasis && !pr.herm && !wt_K &&
                        nufft->prepare_fused_gather(pr.n, d_bls.as<T>(), d_bls.as<T>() + nbls,
                                                    pr.trivial ? nullptr : pr.idx->template as<int>(),
                                                    pr.trivial ? nullptr : pr.flip->template as<signed char>(),
                                                    d_freqs.as<double>() + fa, nfg, tpol, obase,
                                                    (int64_t)nt * per_tf, 1, pol_off, targets_serial,
                                                    mate ? obase + per_tf : nullptr);
                    size_t e4 = ev_begin(TM_FFT, ls);
                    {
                        RoctxRange rr("fft");
                        nufft->fft(ntrans, mate);
                    }
                    ev_end(e4, ls);
                    st[3] += nufft->fft_traffic_cells() * ntrans * nm;
                    st[12] += nufft->fft_flops() * ntrans * nm;
                    RoctxRange rg("gather");
                    size_t e5 = ev_begin(TM_INTERP, ls);
                    BasisTerm bt{d_coefs.p, d_ant1.as<int>(), d_ant2.as<int>(), pr.bi, pr.bj, nbasis,
                                 (int)freqs.size(), fa};
                    // exact eigenbeam symmetry (reference_compat off): the (l, k) term of an off-diagonal pair of
                    // complex basis beams comes from a second gather at -b (all-real pairs: packed, exact already)
                    const int nparts = nbasis && !reference_compat && pr.bi != pr.bj && !pr.herm ? 2 : 1;
                    const WTerm wterm{kt, wt_zc, wt_zh, (const void *)(d_bls.as<T>() + 2 * nbls)};
                    if (!fused)
                        for (int m = 0; m < nm; ++m)
                            for (int part = 1; part <= nparts; ++part) {
                                bt.part = nparts == 2 ? part : 0;
                                bt.negate = nparts == 2 && part == 2;
                                plan_of(*Ls[m])->interp(pr.n, d_bls.as<T>(), d_bls.as<T>() + nbls,
                                      D > 2 ? d_bls.as<T>() + 2 * nbls : nullptr,
                                      pr.trivial ? nullptr : pr.idx->template as<int>(),
                                      pr.trivial ? nullptr : pr.flip->template as<signed char>(),
                                      d_freqs.as<double>() + fa, nfg, tg, obase + (int64_t)m * per_tf,
                                      (int64_t)nt * per_tf, 1, pol_off, accumulate || kt > 0, nbasis ? &bt : nullptr, pr.herm,
                                      pr.ustart ? pr.ustart->template as<int>() : nullptr, pr.upairs ? pr.nitems : pr.nu,
                                      pr.upairs ? pr.upairs->template as<int>() : nullptr, wt_K ? &wterm : nullptr);
                            }
                    ev_end(e5, ls);
                    st[4] += (double)(pr.upairs ? pr.nitems : pr.ustart ? pr.nu : pr.n) * ntrans * nm * (pr.herm ? 2 : 1);  // footprints gathered: distinct targets; packed transforms are read at s and -s
                    if (!light) {  // (the run's own plan describes the run)
                        st[6] = nufft->geo.d[0].n2;
                        st[7] = nufft->geo.d[1].n2;
                        st[8] = nufft->geo.d[0].na * 65536.0 + nufft->geo.d[1].na;
                        st[13] = D > 2 ? nufft->geo.d[2].n2 : 1;
                        st[14] = D > 2 ? nufft->geo.d[2].na : 1;
                        st[9] = nufft->ker.w;
                    }
                    }  // height terms
                }
            }
            if (pipe) {
                if (!heavy_recorded) FV_HIP(hipEventRecord(L0.heavy_done, ls));
                L0.heavy_pending = true;
            }
            if (dbg_t && tu == t0 && chunk == 0) std::fprintf(stderr, "run: first unit queued %.3f s\n", pin.since());
            if (drain && !pin_started) {
                pin.start(device, out, (size_t)nf, run_bytes, sizeof(cplx<T>) * (size_t)out_fs, shared, true);
                pin_started = true;
            }
            close_time();
        }
        if (nlanes > 1 && !pipe) {  // join: everything queued on the main stream afterwards sees every lane
            for (int li = 1; li < nlanes; ++li) {
                FV_HIP(hipEventRecord(lanes[li].done, lanes[li].stream));
                FV_HIP(hipStreamWaitEvent(stream, lanes[li].done, 0));
            }
        }
        if (!out_on_device) {
            if (drain && pin.wait()) {
                const bool dbg = std::getenv("FFTVIS_HIP_DEBUG_DRAIN") != nullptr;
                const double t_queued = pin.since();
                const size_t early = drained;
                drain_flush(out, dout, nt, nf, per_tf, drain_items, drained, out_fs);
                if (dbg) {
                    FV_HIP(hipStreamSynchronize(stream));
                    std::fprintf(stderr, "drain: run queued %.3f s, pinned %.3f s, kernels done %.3f s, ", t_queued,
                                 pin.t_pinned, pin.since());
                }
                FV_HIP(hipStreamSynchronize(copy_stream));
                FV_HIP(hipStreamSynchronize(stream));
                if (dbg) {
                    std::fprintf(stderr, "copies done %.3f s (%zu of %zu time-step items queued before the end)\n",
                                 pin.since(), early, drain_items.size());
                    std::fprintf(stderr, "drain: time steps finished at [ms after the first]:");
                    for (size_t i = 1; i < drain_items.size(); ++i) {
                        float ms = 0;
                        if (hipEventElapsedTime(&ms, drain_items[0].ev, drain_items[i].ev) == hipSuccess) std::fprintf(stderr, " %.0f", ms);
                    }
                    std::fprintf(stderr, "\n");
                }
            } else {
                copy_block_to_host(out, dout, nf, (int64_t)nt * per_tf, out_fs, false, stream);
                FV_HIP(hipStreamSynchronize(stream));
            }
            if (timing_level) ev_collect();
            check_errors();
        }
    }

    // beam x coherency strengths of one (frequency group, beam pair) for the lane's current sources
    void launch_strengths(Lane &L, const Pair &pr, int fa, int nfg, int64_t M, const int *Mp, hipStream_t on, int wt_k = 0,
                          Nufft3<T> *plan = nullptr) {
        Nufft3<T> *nufft = plan ? plan : L.nufft.get();
        const int D = run_D;
        RoctxRange rr("strengths");
        size_t e2 = ev_begin(TM_STRENGTHS, on);
        StrengthArgs sa{};
        sa.M = M;
        sa.nfg = nfg;
        sa.f_first = fa;
        sa.nfreq = nfreq_cat;
        sa.polarized = polarized;
        sa.pol_sky = pol_sky;
        sa.same_beam = pr.bi == pr.bj;
        sa.herm = pr.herm;
        sa.dim = D;
        sa.w = nufft->ker.w;
        for (int d = 0; d < 3; ++d) {
            sa.h[d] = nufft->geo.d[d].h;
            sa.btc[d] = d < D ? pr.box_c()[d] : 0.0;
            sa.na[d] = d < D ? nufft->geo.d[d].na : 1;
        }
        sa.bi = desc(pr.bi);
        sa.bj = desc(pr.bj);
        sa.wt_k = wt_K ? wt_k : 0;
        sa.wt_zc = wt_zc;
        sa.wt_inv = wt_zh > 0 ? 1.0 / wt_zh : 0.0;
        cplx<T> *cs = nufft->strengths_buffer(nfg * (pr.herm ? 2 : tpol));
        hipLaunchKernelGGL((beam_order == 3 ? k_strengths<T, 3> : beam_order == 1 ? k_strengths<T, 1> : k_strengths<T, 0>),
                           dim3(cdiv((int64_t)M * nfg, 256)), dim3(256), 0, on, sa, Mp,
                           nufft->perm.template as<int>(), L.d_srcidx.template as<int>(),
                           L.d_az.template as<T>(), L.d_za.template as<T>(), d_flux.p, d_freqs.as<double>(),
                           nufft->i0s.template as<int>(), nufft->fs.template as<T>(), cs,
                           (const T *)(L.d_xyz.template as<T>() + 2 * M));
        ev_end(e2, on);
    }

    BeamDesc desc(int b) const {
        FV_REQUIRE(b >= 0 && b < (int)beams.size(), "beam pair refers to a missing beam");
        const Beam &bm = beams[b];
        BeamDesc d{};
        d.kind = bm.kind;
        d.order = beam_order;
        d.diameter = bm.diameter;
        for (int i = 0; i < 8; ++i) d.js[i] = bm.js[i];
        d.ps = bm.ps;
        d.table = bm.table ? bm.table->p : nullptr;
        d.nfreq_tab = bm.nfreq_tab;
        d.nza = bm.nza;
        d.naz = bm.naz;
        d.za_max = bm.za_max;
        return d;
    }

    // Called where the host has just synchronised with the main stream: a run that met bad input fails
    // here instead of returning finite, wrong visibilities (finufft rejects such points up front).
    void check_errors() {
        int e[4] = {0, 0, 0, 0};
        FV_HIP(hipMemcpyAsync(e, d_err.p, sizeof(e), hipMemcpyDeviceToHost, stream));
        FV_HIP(hipStreamSynchronize(stream));
        if (!e[0] && !e[1] && !e[2] && !e[3]) return;
        FV_HIP(hipMemsetAsync(d_err.p, 0, sizeof(e), stream));
        if (e[3])
            throw Error(FV_ERR_INTERNAL, "the gather met " + std::to_string(e[3]) + " transform columns its column plan had left "
                                         "out: the visibilities of this run are invalid (FFTVIS_HIP_NO_COLUMN_PLAN=1 avoids the plan)");
        if (e[2])
            throw Error(FV_ERR_ARG, "more sources above the horizon than source_buffer allows (" +
                                        std::to_string(e[2]) + " did not fit): increase source_buffer");
        if (e[1])
            throw Error(FV_ERR_INTERNAL, "type-1 entry buffers overflowed (" + std::to_string(e[1]) +
                                             " entries dropped): the visibilities of this run are invalid");
        throw Error(FV_ERR_ARG, std::to_string(e[0]) +
                                    " source positions were NaN or outside the unit sphere's box (non-unit "
                                    "coord_mgr vectors?): the visibilities of this run are invalid");
    }
    void sync() override {
        FV_HIP(hipSetDevice(device));
        FV_HIP(hipStreamSynchronize(stream));
        if (timing_level) ev_collect();
        check_errors();
    }
    void reserve_mhist(size_t bytes) {
        // growing the buffer frees the counts earlier device-output runs left in it for stats(): fold them first
        if (bytes > d_mhist.cap) fold_mhist();
        d_mhist.reserve(bytes);
    }
    void fold_mhist() {
        // above-horizon counts were left on the device during run(); fold them into the statistics
        if (!mhist_log.empty()) {
            FV_HIP(hipSetDevice(device));
            std::vector<int> mh(d_mhist.cap / sizeof(int));
            FV_HIP(hipMemcpyAsync(mh.data(), d_mhist.p, sizeof(int) * mh.size(), hipMemcpyDeviceToHost, stream));
            FV_HIP(hipStreamSynchronize(stream));
            for (const auto &e : mhist_log) {
                if (e.first < 0 || e.first >= (int)mh.size()) continue;
                st[5] += mh[e.first];
                st[2] += (double)mh[e.first] * e.second;
                st[11] = std::max(st[11], (double)mh[e.first]);
            }
            mhist_log.clear();
        }
    }
    void stats(double *v, int n) override {
        fold_mhist();
        for (int i = 0; i < n && i < 24; ++i) v[i] = st[i];
    }
    void reset_stats() override {
        for (double &x : st) x = 0;
        for (double &x : tm) x = 0;
        spread_timed = 0;
        mhist_log.clear();
        ev_used = 0;
    }
    void enable_timing(int level) override { timing_level = level; }
    void timing(double *ms, int n) override {
        for (int i = 0; i < n && i < TM_COUNT; ++i) ms[i] = tm[i];
        if (n > TM_COUNT) ms[TM_COUNT] = spread_timed;
    }
};

}  // namespace fv
