// fv_nufft.h -- type-3 NUFFT on MI355X: bin-sort -> LDS-tiled spread -> rocFFT -> gather.
//
//   f[t][k] = sum_j c[j][t] exp(+i s_k . x_j)      (finufft type-3 convention, isign = +1;
//                                                   reference call sites src/fftvis/cpu/nufft.py:48-59,105-118)
//
// Design (DESIGN.md "Kernels"):
//  * All ntrans strength vectors share the source points; in the simulator a "trans" is a
//    (frequency, polarisation-product) pair, and each *frequency group* has its own target set
//    scale[g] * (sign_k * b_k), so one spread + one batched FFT serves a whole block of frequencies.
//  * Fine grid is stored CENTRED (cell index = mode + n2/2): sources occupy the middle n1 cells,
//    targets read the middle n2/sigma cells, nothing ever wraps.  The (-1)^index factors this costs
//    are folded into the deconvolution table (input side) and the gather weights (output side).
//  * Spread is output-driven: one workgroup owns one 32x32 tile of the fine grid for TC transforms,
//    accumulates all sources whose footprint touches it in LDS (ds_add_f64 / ds_add_f32), then
//    writes every cell of the tile exactly once (zeros included, deconvolution applied) with 512-B
//    row segments.  No global atomics, no separate memset pass.
#pragma once

#include "fv_eskernel.h"

#include <map>
#include <tuple>

namespace fv {

constexpr int TILE = 32;       // fine-grid tile edge (cells) owned by one spread workgroup
constexpr int GROUP = 16;      // lanes cooperating on one source / one target (>= MAX_W)
constexpr int SPREAD_THREADS = 256;
constexpr int INTERP_THREADS = 256;

struct DimGeom {
    double xc = 0, X = 0;    // source-coordinate centre / half-width
    double btc = 0, B = 0;   // base-target centre / half-width (before frequency scaling)
    double S = 0;            // scale_max * B
    double h = 1;            // x-space grid spacing: xi = (x - xc) / h
    int n1 = 2, n2 = 2;      // active region / FFT length
};

struct Geom {
    int dim = 2;
    DimGeom d[3];
    int ntile[3] = {1, 1, 1};
    int64_t ncell() const { return (int64_t)d[0].n2 * d[1].n2 * (dim > 2 ? d[2].n2 : 1); }
    int ntiles() const { return ntile[0] * ntile[1] * ntile[2]; }
};

inline void set_dim_geom(DimGeom &g, double sigma, int w, double scale_max) {
    g.S = std::fabs(scale_max) * g.B;
    double Xs = g.X, Ss = g.S;
    if (Xs == 0) {
        if (Ss == 0) {
            Xs = 1.0;
            Ss = 1.0;
        } else
            Xs = std::max(Xs, 1.0 / Ss);
    } else
        Ss = std::max(Ss, 1.0 / Xs);
    int n1 = (int)std::ceil(2.0 * sigma * Ss * Xs / M_PI + w + 1);
    n1 += n1 % 2;
    g.n1 = n1;
    g.n2 = next235even((int)std::ceil(sigma * n1));
    g.h = M_PI / (sigma * Ss);
}

// ---------------------------------------------------------------------------------------------
// Device kernels
// ---------------------------------------------------------------------------------------------

struct BinArgs {
    double xc[3], invh[3];
    int n2[3], ntile[3];
    int w, dim;
};

// Footprint start cell i0 = ceil(p - w/2) and first kernel argument f = i0 - p, per dimension,
// plus the tile the footprint's middle cell falls in.  Positions are formed in fp64 and split into
// (int cell, T offset) so that fp32 runs keep sub-cell accuracy on multi-thousand-cell grids.
template <typename T>
__global__ void k_bin_count(int64_t M, const T *__restrict__ x, const T *__restrict__ y,
                            const T *__restrict__ z, BinArgs a, int *__restrict__ i0u,
                            T *__restrict__ fu, int *__restrict__ tile_of, int *__restrict__ counts,
                            int *__restrict__ n_oob) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const T *src[3] = {x, y, z};
    int tl[3] = {0, 0, 0};
    bool oob = false;
    for (int d = 0; d < a.dim; ++d) {
        double p = ((double)src[d][j] - a.xc[d]) * a.invh[d] + 0.5 * a.n2[d];
        int i0 = (int)ceil(p - 0.5 * a.w);
        if (i0 < 0 || i0 + a.w > a.n2[d] || !(p == p)) {  // outside the planned box: never write OOB
            oob = true;
            i0 = max(0, min(a.n2[d] - a.w, i0));
            if (!(p == p)) p = i0 + 0.5 * a.w;
        }
        i0u[(int64_t)d * M + j] = i0;
        fu[(int64_t)d * M + j] = (T)((double)i0 - p);
        tl[d] = (i0 + a.w / 2) / TILE;
    }
    if (oob) atomicAdd(n_oob, 1);
    int t = (tl[2] * a.ntile[1] + tl[1]) * a.ntile[0] + tl[0];
    tile_of[j] = t;
    atomicAdd(&counts[t], 1);
}

// Exclusive scan of n ints by one 1024-thread workgroup; out has n + 1 entries.
__global__ void k_exclusive_scan(const int *__restrict__ in, int *__restrict__ out, int n) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + threadIdx.x;
        int v = i < n ? in[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n) out[i] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

template <typename T>
__global__ void k_bin_scatter(int64_t M, int dim, const int *__restrict__ i0u,
                              const T *__restrict__ fu, const int *__restrict__ tile_of,
                              const int *__restrict__ bin_start, int *__restrict__ cursor,
                              int *__restrict__ i0s, T *__restrict__ fs, int *__restrict__ perm) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    int t = tile_of[j];
    int pos = bin_start[t] + atomicAdd(&cursor[t], 1);
    for (int d = 0; d < dim; ++d) {
        i0s[(int64_t)d * M + pos] = i0u[(int64_t)d * M + j];
        fs[(int64_t)d * M + pos] = fu[(int64_t)d * M + j];
    }
    perm[pos] = (int)j;
}

// Deconvolution table for one dimension of the fine grid, centring sign folded in:
//   tab[i] = (-1)^i / psi_hat(2 pi (i - n2/2) / n2)   for |i - n2/2| <= n1/2, else 0.
template <typename T>
__global__ void k_deconv_table(int n1, int n2, KerParams ker, T *__restrict__ tab) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    int m = i - n2 / 2;
    double v = 0.0;
    if (abs(m) <= n1 / 2) {
        v = 1.0 / es_hat(ker, 2.0 * M_PI * (double)m / (double)n2);
        if (i & 1) v = -v;
    }
    tab[i] = (T)v;
}

// Gather user-order strengths (ntrans, M) into sorted order [M][ntrans] and apply the type-3
// pre-phase exp(i s_c(g) . x'_j); x'_j is rebuilt from the sorted grid coordinates.
template <typename T>
__global__ void k_load_strengths(int64_t M, int ntrans, int tpol, int dim,
                                 const cplx<T> *__restrict__ cin, const int *__restrict__ perm,
                                 const int *__restrict__ i0s, const T *__restrict__ fs, int w,
                                 double h0, double h1, double h2, int n20, int n21, int n22,
                                 double btc0, double btc1, double btc2,
                                 const double *__restrict__ scale, cplx<T> *__restrict__ cs) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M) return;
    const double h[3] = {h0, h1, h2}, btc[3] = {btc0, btc1, btc2};
    const int n2[3] = {n20, n21, n22};
    double dot = 0.0;  // btc . x'
    for (int d = 0; d < dim; ++d) {
        double pos = (double)i0s[(int64_t)d * M + p] - (double)fs[(int64_t)d * M + p];
        dot += btc[d] * (pos - 0.5 * n2[d]) * h[d];
    }
    int j = perm[p];
    for (int t = 0; t < ntrans; ++t) {
        cplx<T> c = cin[(int64_t)t * M + j];
        if (dot != 0.0) {
            double sn, cs_;
            sincos(scale[t / tpol] * dot, &sn, &cs_);
            c = cmul(c, cplx<T>{(T)cs_, (T)sn});
        }
        cs[p * ntrans + t] = c;
    }
}

// --- 2-D spread -------------------------------------------------------------------------------
// grid (ntile_x, ntile_y, ceil(ntrans / TC)); 256 threads = 16 source-groups of 16 lanes.
// Lane g of a group owns footprint column g; the group walks the w footprint rows.
template <typename T, int TC>
__global__ __launch_bounds__(SPREAD_THREADS) void k_spread2d(
    int64_t M, const int *__restrict__ i0s, const T *__restrict__ fs,
    const int *__restrict__ bin_start, const cplx<T> *__restrict__ cs, int ntrans,
    const T *__restrict__ decx, const T *__restrict__ decy, cplx<T> *__restrict__ grid, int n2x,
    int n2y, int ntx, int nty, int w, T beta, T c4) {
    __shared__ T acc[2 * TC * TILE * TILE];  // [re|im][q][row][col]
    const int tid = threadIdx.x;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int t0 = blockIdx.z * TC;
    const int x0 = bx * TILE, y0 = by * TILE;
    for (int i = tid; i < 2 * TC * TILE * TILE; i += SPREAD_THREADS) acc[i] = T(0);
    __syncthreads();

    const int g = tid & (GROUP - 1);       // lane within group = footprint column
    const int grp = tid / GROUP;           // group within workgroup
    const int lane_base = (tid & 63) & ~(GROUP - 1);
    const int *i0x = i0s, *i0y = i0s + M;
    const T *fx = fs, *fy = fs + M;

    for (int dy = -1; dy <= 1; ++dy) {
        int nby = by + dy;
        if (nby < 0 || nby >= nty) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            int nbx = bx + dx;
            if (nbx < 0 || nbx >= ntx) continue;
            const int b = nby * ntx + nbx;
            const int sb = bin_start[b], se = bin_start[b + 1];
            for (int s = sb + grp; s < se; s += SPREAD_THREADS / GROUP) {
                const int relx = i0x[s] - x0, rely = i0y[s] - y0;
                if (relx + w <= 0 || relx >= TILE || rely + w <= 0 || rely >= TILE) continue;
                const T kx = g < w ? es_eval<T>(fx[s] + (T)g, beta, c4) : T(0);
                const T kyv = g < w ? es_eval<T>(fy[s] + (T)g, beta, c4) : T(0);
                T ky[MAX_W];
#pragma unroll
                for (int r = 0; r < MAX_W; ++r) ky[r] = __shfl(kyv, lane_base + r, 64);
                const int col = relx + g;
                const bool colok = g < w && col >= 0 && col < TILE;
#pragma unroll
                for (int q = 0; q < TC; ++q) {
                    if (t0 + q >= ntrans) break;
                    const cplx<T> cv = cs[(int64_t)s * ntrans + t0 + q];
                    const T vr = cv.re * kx, vi = cv.im * kx;
                    T *are = acc + q * TILE * TILE;
                    T *aim = acc + (TC + q) * TILE * TILE;
#pragma unroll
                    for (int r = 0; r < MAX_W; ++r) {
                        const int row = rely + r;
                        if (r < w && colok && row >= 0 && row < TILE) {
                            atomicAdd(&are[row * TILE + col], vr * ky[r]);
                            atomicAdd(&aim[row * TILE + col], vi * ky[r]);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // write-out: every cell of the tile once, deconvolved; a wave covers 2 rows x 512 B.
    const int col = tid & (TILE - 1);
    const int gx = x0 + col;
    if (gx < n2x) {
        const T dxv = decx[gx];
        for (int q = 0; q < TC; ++q) {
            if (t0 + q >= ntrans) break;
            const T *are = acc + q * TILE * TILE;
            const T *aim = acc + (TC + q) * TILE * TILE;
            cplx<T> *plane = grid + (int64_t)(t0 + q) * n2y * n2x;
            for (int row = tid / TILE; row < TILE; row += SPREAD_THREADS / TILE) {
                const int gy = y0 + row;
                if (gy >= n2y) break;
                const T f = dxv * decy[gy];
                plane[(int64_t)gy * n2x + gx] = {are[row * TILE + col] * f, aim[row * TILE + col] * f};
            }
        }
    }
}

// --- 2-D gather (interp) ------------------------------------------------------------------------
struct InterpArgs {
    int dim, w, tpol, nfg;        // tpol transforms per frequency group, nfg groups
    int n2[3];
    double h[3];                  // theta = h * s'
    double btc[3], xc[3];
    double sign;                  // prod_d (-1)^(n2_d / 2)
    int64_t out_fg_stride;        // output element strides
    int64_t out_k_stride;
    int64_t out_pol_off[16];      // offset of polarisation product r (r < tpol <= 16); beyond: r * out_pol_off[1]
    int accumulate;               // out += instead of out =
};

template <typename T>
__global__ __launch_bounds__(INTERP_THREADS) void k_interp2d(
    const cplx<T> *__restrict__ grid, int64_t N, const T *__restrict__ btx,
    const T *__restrict__ bty, const int *__restrict__ bl_idx, const signed char *__restrict__ flip,
    const double *__restrict__ scale, InterpArgs a, KerParams ker, cplx<T> *__restrict__ out) {
    const int tid = threadIdx.x;
    const int g = tid & (GROUP - 1);
    const int lane_base = (tid & 63) & ~(GROUP - 1);
    const int64_t item = (int64_t)blockIdx.x * (INTERP_THREADS / GROUP) + tid / GROUP;
    if (item >= N * a.nfg) return;  // whole group exits together
    const int fg = (int)(item / N);
    const int64_t kl = item % N;
    const int64_t k = bl_idx ? bl_idx[kl] : kl;
    const double sg = (flip && flip[kl]) ? -1.0 : 1.0;
    const double sc = scale[fg];
    const double sx = sc * sg * (double)btx[k], sy = sc * sg * (double)bty[k];  // actual target
    const double spx = sx - sc * a.btc[0], spy = sy - sc * a.btc[1];          // s' = s - s_c
    const double thx = a.h[0] * spx, thy = a.h[1] * spy;
    const double ex = thx * a.n2[0] * (0.5 / M_PI) + 0.5 * a.n2[0];
    const double ey = thy * a.n2[1] * (0.5 / M_PI) + 0.5 * a.n2[1];
    const int w = a.w;
    int j0x = (int)ceil(ex - 0.5 * w), j0y = (int)ceil(ey - 0.5 * w);
    j0x = max(0, min(a.n2[0] - w, j0x));
    j0y = max(0, min(a.n2[1] - w, j0y));
    const T beta = (T)ker.beta, c4 = (T)ker.c;
    T kx = g < w ? es_eval<T>((T)((double)(j0x + g) - ex), beta, c4) : T(0);
    if ((j0x + g) & 1) kx = -kx;
    T kyv = g < w ? es_eval<T>((T)((double)(j0y + g) - ey), beta, c4) : T(0);
    if ((j0y + g) & 1) kyv = -kyv;
    T ky[MAX_W];
#pragma unroll
    for (int r = 0; r < MAX_W; ++r) ky[r] = __shfl(kyv, lane_base + r, 64);

    // psi_1_hat at theta_x, theta_y: quadrature nodes split over the 16 lanes
    double hx = 0.0, hy = 0.0;
    for (int q = g; q < ker.nq; q += GROUP) {
        hx += ker.glf[q] * cos(thx * ker.glz[q]);
        hy += ker.glf[q] * cos(thy * ker.glz[q]);
    }
#pragma unroll
    for (int off = GROUP / 2; off > 0; off >>= 1) {
        hx += __shfl_xor(hx, off, 64);
        hy += __shfl_xor(hy, off, 64);
    }
    double pr = a.sign / (hx * hy), pi_ = 0.0;
    const double ph = sx * a.xc[0] + sy * a.xc[1];  // post-phase exp(i s . x_c)
    if (ph != 0.0) {
        double sn, cs;
        sincos(ph, &sn, &cs);
        pi_ = pr * sn;
        pr = pr * cs;
    }

    const int64_t plane_sz = (int64_t)a.n2[0] * a.n2[1];
    const int gcol = min(j0x + g, a.n2[0] - 1);
    for (int r = 0; r < a.tpol; ++r) {
        const cplx<T> *plane = grid + ((int64_t)fg * a.tpol + r) * plane_sz + gcol;
        T sr = T(0), si = T(0);
#pragma unroll
        for (int rr = 0; rr < MAX_W; ++rr) {
            if (rr < w) {
                const cplx<T> v = plane[(int64_t)(j0y + rr) * a.n2[0]];
                sr += v.re * ky[rr];
                si += v.im * ky[rr];
            }
        }
        sr *= kx;
        si *= kx;
#pragma unroll
        for (int off = GROUP / 2; off > 0; off >>= 1) {
            sr += __shfl_xor(sr, off, 64);
            si += __shfl_xor(si, off, 64);
        }
        if (g == 0) {
            double vr = (double)sr * pr - (double)si * pi_;
            double vi = (double)sr * pi_ + (double)si * pr;
            if (sg < 0) vi = -vi;  // conj for flipped baselines (cpu_simulate.py:298)
            const int64_t po = r < 16 ? a.out_pol_off[r] : (int64_t)r * a.out_pol_off[1];
            cplx<T> *o = out + (int64_t)fg * a.out_fg_stride + po + k * a.out_k_stride;
            if (a.accumulate) {
                o->re += (T)vr;
                o->im += (T)vi;
            } else {
                *o = {(T)vr, (T)vi};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host-side plan
// ---------------------------------------------------------------------------------------------
template <typename T>
struct RocfftPrec;
template <>
struct RocfftPrec<double> {
    static constexpr rocfft_precision v = rocfft_precision_double;
};
template <>
struct RocfftPrec<float> {
    static constexpr rocfft_precision v = rocfft_precision_single;
};

struct FftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    size_t work_bytes = 0;
};

void ensure_rocfft();

template <typename T>
class Nufft3 {
   public:
    int dim;
    double eps, sigma;
    KerParams ker;
    Geom geo;
    hipStream_t stream;
    int64_t M = 0;  // sources currently binned
    int64_t geom_serial = 0;  // bumps whenever the source->cell mapping changes

    // device state
    DevBuf i0u, fu, tile_of, counts, cursor, bin_start, i0s, fs, perm, oob;
    DevBuf dec[3];
    DevBuf grid, work;
    DevBuf strengths;  // [M][ntrans] sorted order
    std::map<std::tuple<int, int, int, int>, FftPlan> fft_cache;
    int64_t stat_spread_cells = 0;  // fine-grid cells written per trans by the last spread

    Nufft3(int dim_, double eps_, double sigma_, hipStream_t s, int w_override = 0)
        : dim(dim_), eps(eps_), sigma(sigma_), stream(s) {
        FV_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
        FV_REQUIRE(sigma == 2.0 || sigma == 1.25, "upsample factor must be 2 or 1.25");
        FV_REQUIRE(eps > 0 && eps < 1, "eps must be in (0, 1)");
        ker = make_kernel(eps, sigma, w_override);
        geo.dim = dim;
        ensure_rocfft();
    }
    ~Nufft3() {
        for (auto &kv : fft_cache) {
            if (kv.second.info) rocfft_execution_info_destroy(kv.second.info);
            if (kv.second.plan) rocfft_plan_destroy(kv.second.plan);
        }
    }

    // Bounds -> grid sizes, deconvolution tables.
    void set_geometry(const double *xc, const double *X, const double *btc, const double *B,
                      double scale_max) {
        Geom old = geo;
        const bool first = geom_serial == 0;
        for (int d = 0; d < dim; ++d) {
            geo.d[d].xc = xc[d];
            geo.d[d].X = X[d];
            geo.d[d].btc = btc[d];
            geo.d[d].B = B[d];
            set_dim_geom(geo.d[d], sigma, ker.w, scale_max);
            geo.ntile[d] = (int)cdiv(geo.d[d].n2, TILE);
        }
        for (int d = 0; d < dim; ++d) {
            if (old.d[d].n1 == geo.d[d].n1 && old.d[d].n2 == geo.d[d].n2 && dec[d].p) continue;
            dec[d].reserve(sizeof(T) * geo.d[d].n2);
            hipLaunchKernelGGL(k_deconv_table<T>, dim3(cdiv(geo.d[d].n2, 256)), dim3(256), 0,
                               stream, geo.d[d].n1, geo.d[d].n2, ker, dec[d].as<T>());
        }
        bool changed = first;
        for (int d = 0; d < dim; ++d)
            if (old.d[d].xc != geo.d[d].xc || old.d[d].h != geo.d[d].h || old.d[d].n2 != geo.d[d].n2)
                changed = true;
        if (changed) {
            ++geom_serial;
            M = 0;
        }
    }

    // Bin-sort the sources for the current geometry (device pointers, length M each).
    void set_sources(int64_t M_, const T *x, const T *y, const T *z) {
        M = M_;
        const int nt = geo.ntiles();
        i0u.reserve(sizeof(int) * 3 * std::max<int64_t>(M, 1));
        fu.reserve(sizeof(T) * 3 * std::max<int64_t>(M, 1));
        i0s.reserve(sizeof(int) * 3 * std::max<int64_t>(M, 1));
        fs.reserve(sizeof(T) * 3 * std::max<int64_t>(M, 1));
        tile_of.reserve(sizeof(int) * std::max<int64_t>(M, 1));
        perm.reserve(sizeof(int) * std::max<int64_t>(M, 1));
        counts.reserve(sizeof(int) * (nt + 1));
        cursor.reserve(sizeof(int) * (nt + 1));
        bin_start.reserve(sizeof(int) * (nt + 1));
        oob.reserve(sizeof(int));
        FV_HIP(hipMemsetAsync(counts.p, 0, sizeof(int) * (nt + 1), stream));
        FV_HIP(hipMemsetAsync(cursor.p, 0, sizeof(int) * (nt + 1), stream));
        FV_HIP(hipMemsetAsync(oob.p, 0, sizeof(int), stream));
        BinArgs a{};
        a.w = ker.w;
        a.dim = dim;
        for (int d = 0; d < 3; ++d) {
            a.xc[d] = geo.d[d].xc;
            a.invh[d] = 1.0 / geo.d[d].h;
            a.n2[d] = d < dim ? geo.d[d].n2 : 1;
            a.ntile[d] = geo.ntile[d];
        }
        if (M > 0) {
            hipLaunchKernelGGL(k_bin_count<T>, dim3(cdiv(M, 256)), dim3(256), 0, stream, M, x, y,
                               z, a, i0u.as<int>(), fu.as<T>(), tile_of.as<int>(),
                               counts.as<int>(), oob.as<int>());
        }
        hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, stream, counts.as<int>(),
                           bin_start.as<int>(), nt);
        if (M > 0) {
            hipLaunchKernelGGL(k_bin_scatter<T>, dim3(cdiv(M, 256)), dim3(256), 0, stream, M, dim,
                               i0u.as<int>(), fu.as<T>(), tile_of.as<int>(), bin_start.as<int>(),
                               cursor.as<int>(), i0s.as<int>(), fs.as<T>(), perm.as<int>());
        }
    }

    int out_of_box_count() {
        int v = 0;
        FV_HIP(hipMemcpyAsync(&v, oob.p, sizeof(int), hipMemcpyDeviceToHost, stream));
        FV_HIP(hipStreamSynchronize(stream));
        return v;
    }

    cplx<T> *strengths_buffer(int ntrans) {
        strengths.reserve(sizeof(cplx<T>) * std::max<int64_t>(M, 1) * ntrans);
        return strengths.as<cplx<T>>();
    }

    // cin: device (ntrans, M) row-major in the caller's source order.
    void load_strengths(const cplx<T> *cin, int ntrans, int tpol, const double *scale_dev) {
        cplx<T> *cs = strengths_buffer(ntrans);
        if (M == 0) return;
        hipLaunchKernelGGL(k_load_strengths<T>, dim3(cdiv(M, 256)), dim3(256), 0, stream, M, ntrans,
                           tpol, dim, cin, perm.as<int>(), i0s.as<int>(), fs.as<T>(), ker.w,
                           geo.d[0].h, geo.d[1].h, geo.d[2].h, geo.d[0].n2, geo.d[1].n2,
                           dim > 2 ? geo.d[2].n2 : 1, geo.d[0].btc, geo.d[1].btc, geo.d[2].btc,
                           scale_dev, cs);
    }

    FftPlan &fft_plan(int ntrans) {
        auto key = std::make_tuple(geo.d[0].n2, geo.d[1].n2, dim > 2 ? geo.d[2].n2 : 1, ntrans);
        auto it = fft_cache.find(key);
        if (it != fft_cache.end()) return it->second;
        FftPlan fp;
        size_t lengths[3] = {(size_t)geo.d[0].n2, (size_t)geo.d[1].n2, (size_t)geo.d[2].n2};
        FV_ROCFFT(rocfft_plan_create(&fp.plan, rocfft_placement_inplace,
                                     rocfft_transform_type_complex_inverse, RocfftPrec<T>::v,
                                     (size_t)dim, lengths, (size_t)ntrans, nullptr));
        FV_ROCFFT(rocfft_plan_get_work_buffer_size(fp.plan, &fp.work_bytes));
        FV_ROCFFT(rocfft_execution_info_create(&fp.info));
        FV_ROCFFT(rocfft_execution_info_set_stream(fp.info, stream));
        return fft_cache.emplace(key, fp).first->second;
    }

    void spread(int ntrans);
    void fft(int ntrans) {
        FftPlan &fp = fft_plan(ntrans);
        if (fp.work_bytes) {
            work.reserve(fp.work_bytes);
            FV_ROCFFT(rocfft_execution_info_set_work_buffer(fp.info, work.p, fp.work_bytes));
        }
        void *bufs[1] = {grid.p};
        FV_ROCFFT(rocfft_execute(fp.plan, bufs, nullptr, fp.info));
    }
    // Targets: base coordinates bt* (device, indexed by global baseline id), optional subset
    // index list / flip flags of length N, per-group scale (device, nfg doubles).
    void interp(int64_t N, const T *btx, const T *bty, const T *btz, const int *bl_idx,
                const signed char *flip, const double *scale_dev, int nfg, int tpol,
                cplx<T> *out, int64_t out_fg_stride, int64_t out_k_stride,
                const int64_t *out_pol_off, bool accumulate);
};

template <typename T>
void Nufft3<T>::spread(int ntrans) {
    grid.reserve(sizeof(cplx<T>) * geo.ncell() * ntrans);
    FV_REQUIRE(dim == 2, "3-D spread not built yet");
    constexpr int TC = sizeof(T) == 8 ? 4 : 8;  // 64 KiB of LDS accumulators per workgroup
    dim3 g(geo.ntile[0], geo.ntile[1], (unsigned)cdiv(ntrans, TC));
    hipLaunchKernelGGL((k_spread2d<T, TC>), g, dim3(SPREAD_THREADS), 0, stream, M, i0s.as<int>(),
                       fs.as<T>(), bin_start.as<int>(), strengths.as<cplx<T>>(), ntrans,
                       dec[0].as<T>(), dec[1].as<T>(), grid.as<cplx<T>>(), geo.d[0].n2,
                       geo.d[1].n2, geo.ntile[0], geo.ntile[1], ker.w, (T)ker.beta, (T)ker.c);
    stat_spread_cells = geo.ncell();
}

template <typename T>
void Nufft3<T>::interp(int64_t N, const T *btx, const T *bty, const T *btz, const int *bl_idx,
                       const signed char *flip, const double *scale_dev, int nfg, int tpol,
                       cplx<T> *out, int64_t out_fg_stride, int64_t out_k_stride,
                       const int64_t *out_pol_off, bool accumulate) {
    if (N == 0 || nfg == 0) return;
    FV_REQUIRE(dim == 2, "3-D interp not built yet");
    InterpArgs a{};
    a.dim = dim;
    a.w = ker.w;
    a.tpol = tpol;
    a.nfg = nfg;
    a.sign = 1.0;
    for (int d = 0; d < 3; ++d) {
        a.n2[d] = d < dim ? geo.d[d].n2 : 1;
        a.h[d] = geo.d[d].h;
        a.btc[d] = geo.d[d].btc;
        a.xc[d] = geo.d[d].xc;
        if (d < dim && ((geo.d[d].n2 / 2) & 1)) a.sign = -a.sign;
    }
    a.out_fg_stride = out_fg_stride;
    a.out_k_stride = out_k_stride;
    for (int r = 0; r < 16; ++r) a.out_pol_off[r] = out_pol_off ? out_pol_off[r] : 0;
    a.accumulate = accumulate ? 1 : 0;
    const int64_t items = N * nfg;
    hipLaunchKernelGGL(k_interp2d<T>, dim3(cdiv(items, INTERP_THREADS / GROUP)),
                       dim3(INTERP_THREADS), 0, stream, grid.as<cplx<T>>(), N, btx, bty, bl_idx,
                       flip, scale_dev, a, ker, out);
}

}  // namespace fv
