// fv_nufft.h -- type-3 NUFFT on MI355X: bin-sort -> LDS-tiled spread -> pruned FFT -> gather.
//
//   f[t][k] = sum_j c[j][t] exp(+i s_k . x_j)      (finufft type-3 convention, isign = +1;
//                                                   reference call sites src/fftvis/cpu/nufft.py:48-59,105-118)
//
// Design (DESIGN.md "Kernels"):
//  * All ntrans strength vectors share the source points; in the simulator a "trans" is a
//    (frequency, polarisation-product) pair, and each *frequency group* has its own target set
//    scale[g] * (sign_k * b_k), so one spread + one batched FFT serves a whole block of frequencies.
//  * Only the part of the fine grid that can be non-zero is ever materialised.  Per dimension the
//    sources touch `na` ~ n2/sigma cells around mode 0 (buffer A, mode m = index - na/2) and the
//    targets read `no` ~ n2/sigma cells around mode 0 of the transform (mode l = index - no/2).
//    The uniform step  g_l = sum_m b_m exp(+2 pi i m l / n2)  is therefore a PRUNED FFT: row
//    kernels read na inputs, run length-Q power-of-two FFTs (n2 = P*Q, decimation in frequency
//    over the P residues: every (row, residue) is an independent job) with the radix passes held
//    in registers and LDS as the exchange between them (k_rowfft_st; k_rowfft_dif for Q < 512), and
//    write no outputs; the y-pass reads columns directly when they are short (fusing the
//    transpose), a tile transpose sits between the passes otherwise.  HBM traffic ~ 4-6 na^2 cells
//    per transform instead of the >= 8 n2^2 = 32 na^2 of a full-grid library FFT (four passes) plus
//    the zero padding the spread would have to write.  On small 2-D grids the last pass also
//    serves the targets from its LDS tiles (fused gather): no output grid at all.  Rows are centred
//    by entering element ia at the cyclic slot (ia - n_in/2) mod Q, so no output carries a phase.
//  * Spread is a GATHER: one wave owns one 8x8-cell block of A (lane = cell) for TCH transforms.
//    Sources are counting-sorted into 8x8-cell bins of their footprint origin, their 2w kernel
//    weights are evaluated once per (time, geometry); the wave walks the <= 3x3 bins whose
//    footprints can reach its block in chunks of 16 sources staged through registers into its
//    private LDS slice, multiplies the two tabulated weights and accumulates strengths in
//    registers, then writes every cell once (zeros included, deconvolution applied).  No atomics,
//    no memset pass; blocks are processed heaviest-first.  Source-dense 2-D grids use a second lane
//    mapping (lane = block column x channel group, k_spread2d_cg) that reads far less LDS per visit.
//  * spread() and fft() take a `mate`: a second plan of the same geometry (another time step) whose
//    sources / grids the same launches serve through blockIdx.y (gang launches, see fv_sim.h).
#pragma once

#include "fv_eskernel.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <array>
#include <map>
#include <memory>

namespace fv {

// A/B switches for measurements (read once): FFTVIS_HIP_OLD_FFT routes every row FFT through the
// LDS-resident kernel, FFTVIS_HIP_NATURAL_ORDER stores all FFT outputs in natural (interleaved) order.
inline bool debug_switch_old_fft() {
    static const bool v = std::getenv("FFTVIS_HIP_OLD_FFT") != nullptr;
    return v;
}
inline bool debug_switch_natural_order() {
    static const bool v = std::getenv("FFTVIS_HIP_NATURAL_ORDER") != nullptr;
    return v;
}

constexpr int BINLOG = 3;      // sources are binned by footprint origin in 8x8-cell bins
constexpr int GROUP = 16;      // lanes cooperating on one source / one target (>= MAX_W)
constexpr int SPREAD_THREADS = 256;
#ifndef FV_INTERP_THREADS
#define FV_INTERP_THREADS 256
#endif
constexpr int INTERP_THREADS = FV_INTERP_THREADS;
constexpr int FFT_THREADS = 256;
#ifndef FV_FFT_QMAX_LOG
#define FV_FFT_QMAX_LOG 12
#endif
#ifndef FV_FFT_TPR_DIV1
#define FV_FFT_TPR_DIV1 8  // threads per row = Q / this
#endif
constexpr int FFT_QMAX_LOG = FV_FFT_QMAX_LOG;  // LDS row buffer: Q <= 2^this complex (70 KiB fp64 at 4096)

struct DimGeom {
    double xc = 0, X = 0;    // source-coordinate centre / half-width
    double btc = 0, B = 0;   // base-target centre / half-width (before frequency scaling)
    double S = 0;            // scale_max * B
    double h = 1;            // x-space grid spacing: xi = (x - xc) / h
    int n1 = 2;              // cells the sources can touch
    int na = 8;              // n1 rounded up to whole 8-cell bins: extent of buffer A
    int n2 = 64, P = 1, Q = 64, logQ = 6;  // FFT length n2 = P * Q
    int no = 2;              // transform outputs kept (centred on mode 0)
    // Outputs are stored residue-major: output index idx = l + no/2 sits at
    // (idx mod P) * cnt + idx / P of a row of nos = P * cnt slots.  A residue job of the pruned FFT
    // (l = P k' + p) then writes one contiguous run instead of every P-th 16-B element -- the
    // interleaved stores of separate workgroups cost 2.6x the algorithmic write traffic at P = 2.
    // Every consumer (next pass, transpose, gather, mode pick) addresses columns through out_pos().
    // The last dimension (contiguous for the gather, whose lanes read w consecutive outputs) keeps
    // the natural order: rm = false.
    bool rm = true;
    int sP() const { return rm ? P : 1; }
    int cnt() const { return (no + sP() - 1) / sP(); }
    int nos() const { return cnt() * sP(); }
};
__host__ __device__ inline int out_pos(int idx, int P, int cnt) {
    return P == 1 ? idx : (idx % P) * cnt + idx / P;
}

struct Geom {
    int dim = 2;
    DimGeom d[3];
    int nbin[3] = {1, 1, 1};
    int nbins() const { return nbin[0] * nbin[1] * nbin[2]; }
    int64_t cells_a() const { return (int64_t)d[0].na * d[1].na * (dim > 2 ? d[2].na : 1); }
    int64_t cells_o() const { return (int64_t)d[0].no * d[1].no * (dim > 2 ? d[2].no : 1); }
};

// n2 = P * 2^b >= nmin with 16 <= 2^b <= 4096 and P <= 16 (P unbounded at 2^b = 4096), chosen to
// minimise n2 * (1 + 0.03 (P - 1)): HBM traffic grows with n2, per-row work with P.  Then as many
// factors of two as possible move from P into Q.
inline void choose_pq(int nmin, DimGeom &g, bool last_dim = false) {
    double best = 0;
    int bp = 0, bq = 0;
    for (int b = 4; b <= FFT_QMAX_LOG; ++b) {
        const int q = 1 << b;
        const int p = (nmin + q - 1) / q;
        if (p > 16 && b < FFT_QMAX_LOG) continue;
        int pp = p, bb = b;
        while (pp % 2 == 0 && bb < FFT_QMAX_LOG) {
            pp /= 2;
            ++bb;
        }
        // every extra residue re-reads the row (from L2) and re-applies the input twiddles: measured on C3's
        // mix of grid sizes with the sweep-wise fold of k_rowfft_st<.., FOLD>: 0.05-0.12 per extra residue
        // beats 0.25 by 2 % (10240 = 5 x 2048 instead of 12288 = 3 x 4096 for the widest grids)
        static const double pen0 = std::getenv("FFTVIS_HIP_PQ_PENALTY") ? std::atof(std::getenv("FFTVIS_HIP_PQ_PENALTY")) : 0.12;
        // the last dimension (column passes: 8 columns of Q <= 1024 or 4 of Q = 2048 per workgroup) takes its own penalty
        static const double pen1 = std::getenv("FFTVIS_HIP_PQ_PENALTY_LAST") ? std::atof(std::getenv("FFTVIS_HIP_PQ_PENALTY_LAST")) : pen0;
        const double pen = last_dim ? pen1 : pen0;
        const double cost = (double)p * q * (1.0 + pen * (pp - 1));
        if (best == 0 || cost < best) {
            best = cost;
            bp = pp;
            bq = bb;
        }
    }
    g.P = bp;
    g.Q = 1 << bq;
    g.logQ = bq;
    g.n2 = bp << bq;
}

// Dimensions after the first are transformed along columns of the previous pass's output.  The column
// kernel holds 4 columns of Q <= 2048 per workgroup; a Q = 4096 column needs a tile transpose and a row
// pass instead.  Same n2, one factor of two moved back from Q into P (4 x 2048 instead of 2 x 4096): the
// fold costs P n_in complex FMAs more, the radix passes one level less, and the transpose goes away.
inline void cap_column_q(DimGeom &g) {
    static const int cap = std::getenv("FFTVIS_HIP_COL_QCAP") ? std::atoi(std::getenv("FFTVIS_HIP_COL_QCAP")) : 11;
    while (g.logQ > cap && g.logQ > 4 && 2 * g.P <= 16) {
        g.P *= 2;
        g.Q /= 2;
        --g.logQ;
    }
}

// How much of the rounding slack of n2 goes into a finer source grid (0 = none, 1 = all of it): see set_dim_geom.
inline double geom_slack_share() {
    const char *e = std::getenv("FFTVIS_HIP_GRID_SLACK");  // read per geometry: tests flip it
    return e ? std::min(1.0, std::max(0.0, std::atof(e))) : 1.0;
}

// last_dim: the dimension the gather reads contiguously (transformed by the last pass).
inline void set_dim_geom(DimGeom &g, double sigma, int w, double scale_max, bool last_dim = true, double slack_share = -1.0) {
    if (slack_share < 0.0) slack_share = geom_slack_share();
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
    // the gather's footprints reach |eta| <= n2 / (2 sigma) + w / 2 and must stay inside the n2 outputs
    // (no periodic wrap there): n2 (1 - 1 / sigma) >= w + 4.  Only tiny grids at sigma = 1.25 are affected
    // (they lost up to 250 eps at tight tolerances before).
    const int nwrap = (int)std::ceil((w + 4) / (1.0 - 1.0 / sigma));
    choose_pq(std::max({(int)cdiv(n1, 1 << BINLOG) << BINLOG, (int)std::ceil(sigma * n1), nwrap}), g, last_dim);
    // n2 = P 2^b is rounded up, by up to a third (6144, 8192, 10240 ...), and the transform itself only needs
    // n2 >= sigma n1.  The slack goes into a FINER source grid: spacing h = pi / (so S) with so >= sigma -- the
    // outer step more oversampled than asked, never less accurate -- grown until sigma n1(so) reaches n2.  The
    // sources then touch more cells (n1 up), but the targets span fewer transform outputs, |eta| <= n2 / (2 so),
    // so every LATER pass has fewer lines to transform and the gather's grid shrinks (C3, x: na 4688 -> 5120, no
    // 5132 -> 4700 at n2 = 10240).  Not in the last dimension, which has no later pass to gain from it, and not under a
    // column plan (Nufft3::arm_columns), which keeps only the columns the targets read however finely they are sampled:
    // the caller then passes slack_share = 0 (Nufft3::grid_slack).
    double so = sigma;
    if (!last_dim && slack_share > 0.0) {
        const double so_max = ((double)g.n2 / sigma - w - 3.0) * M_PI / (2.0 * Ss * Xs);
        if (so_max > sigma) {
            const double so_try = sigma + slack_share * (so_max - sigma);
            int n1s = (int)std::ceil(2.0 * so_try * Ss * Xs / M_PI + w + 1);
            n1s += n1s % 2;
            const int nas = (int)cdiv(n1s, 1 << BINLOG) << BINLOG;
            if (n1s >= n1 && (double)n1s * sigma <= (double)g.n2 && nas <= g.n2) {
                so = so_try;
                n1 = n1s;
            }
        }
    }
    g.n1 = n1;
    g.na = (int)cdiv(n1, 1 << BINLOG) << BINLOG;  // whole source bins
    g.h = M_PI / (so * Ss);
    // targets sit at |eta| <= n2/(2 so) (in transform cells); keep the footprint around them
    g.no = 2 * ((int)std::ceil(0.5 * g.n2 / so) + w / 2 + 2);
    g.no = std::min(g.no, g.n2);
}

// ---------------------------------------------------------------------------------------------
// Device kernels
// ---------------------------------------------------------------------------------------------

struct BinArgs {
    double xc[3], invh[3];
    int na[3], nbin[3];
    int w, dim;
    double r2max;  // > 0 (2-D): the sources lie in the disc x^2 + y^2 <= r2max (Nufft3::disc_radius); outside = out of box
};

// Footprint start cell i0 = ceil(p - w/2) in buffer-A coordinates and first kernel argument
// f = i0 - p, per dimension, plus the 8x8 bin the footprint origin falls in.  Positions are
// formed in fp64 and split into (int cell, T offset) so that fp32 runs keep sub-cell accuracy on
// multi-thousand-cell grids.
template <typename T>
__global__ void k_bin_count(int64_t M, const int *__restrict__ Mp, const T *__restrict__ x,
                            const T *__restrict__ y, const T *__restrict__ z, BinArgs a,
                            int *__restrict__ i0u, T *__restrict__ fu, int *__restrict__ tile_of,
                            int *__restrict__ counts, int *__restrict__ n_oob) {
    // M = capacity (array stride, launch bound); *Mp = live count when it is only known on device
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (Mp ? min((int64_t)*Mp, M) : M)) return;
    const T *src[3] = {x, y, z};
    int tl[3] = {0, 0, 0};
    bool oob = false;
    for (int d = 0; d < a.dim; ++d) {
        double p = ((double)src[d][j] - a.xc[d]) * a.invh[d] + 0.5 * a.na[d];
        int i0 = (int)ceil(p - 0.5 * a.w);
        if (i0 < 0 || i0 + a.w > a.na[d] || !(p == p)) {  // outside the planned box: never write OOB
            oob = true;
            i0 = max(0, min(a.na[d] - a.w, i0));
            if (!(p == p)) p = i0 + 0.5 * a.w;
        }
        i0u[(int64_t)d * M + j] = i0;
        fu[(int64_t)d * M + j] = (T)((double)i0 - p);
        tl[d] = i0 >> BINLOG;
    }
    if (a.r2max > 0.0) {  // blocks outside the disc are neither written by the spread nor read by the x-pass
        const double sx = (double)src[0][j], sy = (double)src[1][j];
        if (!(sx * sx + sy * sy <= a.r2max)) {
            oob = true;
            tl[0] = a.nbin[0] / 2;  // counted as an error; parked in a block that exists
            tl[1] = a.nbin[1] / 2;
            i0u[j] = tl[0] << BINLOG;
            i0u[M + j] = tl[1] << BINLOG;
        }
    }
    if (oob) atomicAdd(n_oob, 1);
    int t = (tl[2] * a.nbin[1] + tl[1]) * a.nbin[0] + tl[0];
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

// Three-phase scan for large n (hundreds of thousands of bins): per-1024-block local scans and
// block totals, a scan of the totals (k_exclusive_scan), then the offsets are added back.
__global__ void k_scan_blocks(const int *__restrict__ in, int *__restrict__ out,
                              int *__restrict__ totals, int n) {
    __shared__ int wsum[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int v = i < n ? in[i] : 0;
    int x = v;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wv; ++k) base += wsum[k];
    if (i < n) out[i] = base + x - v;
    if (threadIdx.x == 1023) totals[blockIdx.x] = base + x;
}

__global__ void k_scan_add(int *__restrict__ out, const int *__restrict__ block_off, int n) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += block_off[blockIdx.x];
    if (i == n - 1 || (n == 0 && i == 0)) out[n] = block_off[gridDim.x];
}

// Counting sort, second half, in two kernels so that the order INSIDE a bin does not depend on the order in
// which atomics retire: (1) every source takes a slot of its bin (atomic cursor) and leaves its id there;
// (2) every slot's source finds its rank among the ids of its bin and moves to that position, where its
// footprint origin, offsets and tabulated kernel weights  kw[(d * M + pos) * w + k] = psi(f_d + k), k < w,
// are written.  The spread sums a cell's contributions in bin-walk order, so with (2) two runs of the same
// transform are bit-identical.  Bins beyond RANK_SORT_MAX sources keep the atomic order (a clustered
// catalog would make the rank search quadratic).
constexpr int RANK_SORT_MAX = 8192;
__global__ void k_bin_scatter_ids(int64_t M, const int *__restrict__ Mp, const int *__restrict__ tile_of,
                                  const int *__restrict__ bin_start, int *__restrict__ cursor,
                                  int *__restrict__ slot_id) {
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (Mp ? min((int64_t)*Mp, M) : M)) return;
    const int t = tile_of[j];
    slot_id[bin_start[t] + atomicAdd(&cursor[t], 1)] = (int)j;
}

template <typename T>
__global__ void k_bin_fill(int64_t M, const int *__restrict__ Mp, int dim, const int *__restrict__ i0u,
                           const T *__restrict__ fu, const int *__restrict__ tile_of,
                           const int *__restrict__ bin_start, const int *__restrict__ slot_id,
                           int *__restrict__ i0s, T *__restrict__ fs, int *__restrict__ perm,
                           T *__restrict__ kw, int w, T beta, T c4) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (Mp ? min((int64_t)*Mp, M) : M)) return;
    const int j = slot_id[q];
    const int t = tile_of[j];
    const int s0 = bin_start[t], s1 = bin_start[t + 1];
    int rank = (int)q - s0;
    if (s1 - s0 > 1 && s1 - s0 <= RANK_SORT_MAX) {
        rank = 0;
        for (int k = s0; k < s1; ++k) rank += slot_id[k] < j;
    }
    const int64_t pos = s0 + rank;
    for (int d = 0; d < dim; ++d) {
        const T f = fu[(int64_t)d * M + j];
        i0s[(int64_t)d * M + pos] = i0u[(int64_t)d * M + j];
        fs[(int64_t)d * M + pos] = f;
        T *row = kw + ((int64_t)d * M + pos) * w;
        for (int k = 0; k < w; ++k) row[k] = es_eval<T>(f + (T)k, beta, c4);
    }
    perm[pos] = j;
}

// Inner-kernel deconvolution table of one dimension in buffer-A coordinates:
//   tab[i] = 1 / psi_hat(2 pi (i - na/2) / n2).
template <typename T>
__global__ void k_deconv_table(int na, int n2, KerParams ker, T *__restrict__ tab) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    const int m = i - na / 2;
    tab[i] = (T)(1.0 / es_hat(ker, 2.0 * M_PI * (double)m / (double)n2));
}

// tw[j] = exp(+2 pi i j / n2), j in [0, n2)  (inverse-sign FFT twiddles, formed in fp64)
template <typename T>
__global__ void k_twiddle_table(int n2, cplx<T> *__restrict__ tw) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n2) return;
    double s, c;
    sincospi(2.0 * (double)j / (double)n2, &s, &c);
    tw[j] = {(T)c, (T)s};
}

// Gather user-order strengths (ntrans, M) into sorted order [M][ntrans] and apply the type-3
// pre-phase exp(i s_c(g) . x'_j); x'_j is rebuilt from the sorted grid coordinates.
template <typename T>
__global__ void k_load_strengths(int64_t M, const int *__restrict__ Mp, int ntrans, int tpol, int dim,
                                 const cplx<T> *__restrict__ cin, const int *__restrict__ perm,
                                 const int *__restrict__ i0s, const T *__restrict__ fs, double h0,
                                 double h1, double h2, int na0, int na1, int na2, double btc0,
                                 double btc1, double btc2, const double *__restrict__ scale,
                                 cplx<T> *__restrict__ cs) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (Mp ? min((int64_t)*Mp, M) : M)) return;
    const double h[3] = {h0, h1, h2}, btc[3] = {btc0, btc1, btc2};
    const int na[3] = {na0, na1, na2};
    double dot = 0.0;  // btc . x'
    for (int d = 0; d < dim; ++d) {
        double pos = (double)i0s[(int64_t)d * M + p] - (double)fs[(int64_t)d * M + p];
        dot += btc[d] * (pos - 0.5 * na[d]) * h[d];
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

// Second source set of a gang launch (grid.y = 2: two time steps on one geometry in one launch).
struct SpreadMate {
    const int *i0s, *bin_start;
    const void *kw, *cs;
    void *grid;
};

// --- 2-D spread (gather) ------------------------------------------------------------------------
// One wave owns one 8x8-cell block of A, aligned with the 8x8 source bins; lane = cell.  A source
// touches cell c iff 0 <= c - i0 < w in both dimensions, so only the bins
// (8 b - w + 1) >> 3 .. b of each dimension can reach block b (2x2 bins for w <= 9, 3x3 up to 16).
// The wave walks those sources in chunks of SPREAD_CHUNK: it first stages the chunk's footprint
// origins, tabulated kernel weights and TCH strengths into its own slice of LDS with bulk coalesced
// loads (many in flight), then every lane accumulates from LDS -- strengths and origins are
// broadcast reads, the two weights are per-lane -- TCH transforms in registers.  Every cell is
// written once (zeros included, deconvolution applied).  grid (ceil(nbx / 4), nby, chunks).
constexpr int SPREAD_CHUNK = 16;

#ifndef FV_SPREAD_MINW
#define FV_SPREAD_MINW 3
#endif
template <typename T, int TCH>
__global__ __launch_bounds__(SPREAD_THREADS, FV_SPREAD_MINW) void k_spread2d(
    int64_t M, const int *__restrict__ i0s_a, const T *__restrict__ kw_a,
    const int *__restrict__ bin_start_a, const cplx<T> *__restrict__ cs_a, int ntrans, int tbegin,
    const T *__restrict__ decx, const T *__restrict__ decy, cplx<T> *__restrict__ grid_a, int nax,
    int nay, int nbx, int w, const int *__restrict__ order, int nchunk, SpreadMate mate) {
    const int *__restrict__ i0s = blockIdx.y ? mate.i0s : i0s_a;
    const T *__restrict__ kw = blockIdx.y ? static_cast<const T *>(mate.kw) : kw_a;
    const int *__restrict__ bin_start = blockIdx.y ? mate.bin_start : bin_start_a;
    const cplx<T> *__restrict__ cs = blockIdx.y ? static_cast<const cplx<T> *>(mate.cs) : cs_a;
    cplx<T> *__restrict__ grid = blockIdx.y ? static_cast<cplx<T> *>(mate.grid) : grid_a;
    __shared__ cplx<T> s_str[SPREAD_THREADS / 64][SPREAD_CHUNK][TCH];
    __shared__ T s_kw[SPREAD_THREADS / 64][SPREAD_CHUNK][2][MAX_W];
    __shared__ int s_i0[SPREAD_THREADS / 64][SPREAD_CHUNK][2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // workgroup id -> (4 blocks of one bin row, transform chunk): `order` lists the 4-block groups
    // by decreasing expected load (see Nufft3::build_block_order), chunks fastest, so that the
    // heavy groups of every chunk start first and the light ones fill the tail
    const int og = order[blockIdx.x / nchunk];
    const int bx = (og & 0xffff) * 4 + wave, by = og >> 16;
    if (bx >= nbx) return;  // wave-uniform
    const int tbase = tbegin + (blockIdx.x % nchunk) * TCH;  // the launcher only issues whole chunks
    const int cx = (bx << BINLOG) + (lane & 7), cy = (by << BINLOG) + (lane >> 3);
    const int *i0x = i0s, *i0y = i0s + M;
    const T *kwx = kw, *kwy = kw + M * w;
    T ar[TCH], ai[TCH];
#pragma unroll
    for (int q = 0; q < TCH; ++q) ar[q] = ai[q] = T(0);
    const int bxl = max((bx << BINLOG) - w + 1, 0) >> BINLOG;
    const int byl = max((by << BINLOG) - w + 1, 0) >> BINLOG;
    // bins bxl .. bx of one bin row are contiguous in the sorted order: up to 3 source ranges
    // (w <= 16), walked as one sequence of chunks
    int rs0[3], rnc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yb = min(byl + r, by);
        const int s0 = bin_start[yb * nbx + bxl], s1 = bin_start[yb * nbx + bx + 1];
        rs0[r] = s0;
        rnc[r] = byl + r <= by ? s1 - s0 : 0;  // sources in the range
    }
    // the sources of the <= 3 bin rows form ONE visit list (rows one after the other) cut into chunks of 16: separate
    // lists made up to three short chunks per block, and the chunks are a serial chain of load round trips
    const int L0 = rnc[0], L01 = rnc[0] + rnc[1], ntot = L01 + rnc[2];
    const int nct = (ntot + SPREAD_CHUNK - 1) / SPREAD_CHUNK;
    auto src_of = [&](int i) -> int {  // visit-list element i -> sorted source index
        return i < L0 ? rs0[0] + i : i < L01 ? rs0[1] + (i - L0) : rs0[2] + (i - L01);
    };
    // a chunk travels global -> registers (requested one chunk ahead, so the loads fly while the
    // previous chunk is accumulated) -> the wave's LDS slice -> broadcast reads
    constexpr int NS = TCH >= 4 ? TCH / 4 : 1;  // strengths per lane: 16 TCH / 64
    constexpr int NW = (SPREAD_CHUNK * MAX_W) / 64;  // weights per lane and dimension (w <= MAX_W)
    cplx<T> ps[NS];
    T pkx[NW], pky[NW];
    int pix = 0, piy = 0;
    int wj[NW], wk[NW];  // weight element e = lane + 64 i of a chunk: slot e / w, tap e % w (the same for every chunk)
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int e = lane + 64 * i;
        wj[i] = e / w;
        wk[i] = e - wj[i] * w;
    }
    auto request = [&](int base, int n) {  // base: first visit-list element of the chunk
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int e = lane + 64 * i;
            ps[i] = {T(0), T(0)};
            if (e < n * TCH) ps[i] = cs[(int64_t)src_of(base + e / TCH) * ntrans + tbase + e % TCH];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            pkx[i] = pky[i] = T(0);
            if (wj[i] < n) {
                const int64_t at = (int64_t)src_of(base + wj[i]) * w + wk[i];
                pkx[i] = kwx[at];
                pky[i] = kwy[at];
            }
        }
        if (lane < n) {
            const int sidx = src_of(base + lane);
            pix = i0x[sidx];
            piy = i0y[sidx];
        }
    };
    int n = 0, base = 0;
    if (nct > 0) {
        n = min(SPREAD_CHUNK, ntot);
        request(0, n);
    }
    for (int c = 0; c < nct; ++c) {
        // ---- registers -> LDS (wave-private slice; same-wave LDS ops stay in order) -----------
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int e = lane + 64 * i;
            if (e < n * TCH) s_str[wave][e / TCH][e % TCH] = ps[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (wj[i] < n) {
                s_kw[wave][wj[i]][0][wk[i]] = pkx[i];
                s_kw[wave][wj[i]][1][wk[i]] = pky[i];
            }
        }
        if (lane < n) {
            s_i0[wave][lane][0] = pix;
            s_i0[wave][lane][1] = piy;
        }
        const int ncur = n;
        if (c + 1 < nct) {
            base = (c + 1) * SPREAD_CHUNK;
            n = min(SPREAD_CHUNK, ntot - base);
            request(base, n);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- accumulate from LDS --------------------------------------------------------------
        for (int j = 0; j < ncur; ++j) {
            if (w > 9) {  // the bins two back reach this block only with the far end of their footprints
                const int sx = __builtin_amdgcn_readfirstlane(s_i0[wave][j][0]);
                const int sy = __builtin_amdgcn_readfirstlane(s_i0[wave][j][1]);
                if (sx + w <= (bx << BINLOG) || sy + w <= (by << BINLOG)) continue;  // wave-uniform
            }
            const int dx = cx - s_i0[wave][j][0], dy = cy - s_i0[wave][j][1];
            T wt = T(0);
            if ((unsigned)dx < (unsigned)w && (unsigned)dy < (unsigned)w)
                wt = s_kw[wave][j][0][dx] * s_kw[wave][j][1][dy];
#pragma unroll
            for (int q = 0; q < TCH; ++q) {
                const cplx<T> cv = s_str[wave][j][q];
                ar[q] += cv.re * wt;
                ai[q] += cv.im * wt;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the slice is rewritten by the next chunk
    }
    const T f = decx[cx] * decy[cy];
    const int64_t plane = (int64_t)nay * nax;
    cplx<T> *o = grid + (int64_t)tbase * plane + (int64_t)cy * nax + cx;
#pragma unroll
    for (int q = 0; q < TCH; ++q) o[q * plane] = {ar[q] * f, ai[q] * f};
}

// --- 2-D spread, channel-group lanes (TCH >= 8) ----------------------------------------------------
// Same block / bin / chunk walk as k_spread2d, other lane mapping: lane = (x cell 0..7, channel group
// 0..7) and a lane accumulates CPL = TCH / 8 transforms for the 8 cells of its block column.  In
// k_spread2d every lane reads all TCH strengths of a source from LDS (256 B per lane and source
// visit, and LDS returns bytes per lane whether or not the address is shared: that was the kernel's
// bound once the sources were staged); here a lane reads its own CPL strengths, one x weight and
// the 8 y weights of its column (<= 112 B).  The y-weight rows are zero-padded on both sides so
// that those reads need no range checks (dy = 8 by + k - i0y lies in [-7, w + 13]); the footprint
// origins come out of the staging registers with v_readlane (wave-uniform, no LDS round trip).
// Skipping the rows (or half blocks) a footprint misses with scalar branches was measured slower.
// Pays once blocks hold a few sources each (C2: 33.5 -> 31 us per launch, 4x the catalog: 96 -> 78 us);
// on sparse grids (C3: < 1 source per block) the lane-per-cell kernel stays ahead, see launch_spread.
constexpr int SPREAD_KWP = MAX_W + 24;  // padded y-weight row: 8 leading zeros, w weights, zeros
template <typename T, int TCH>
__global__ __launch_bounds__(SPREAD_THREADS, FV_SPREAD_MINW) void k_spread2d_cg(
    int64_t M, const int *__restrict__ i0s_a, const T *__restrict__ kw_a,
    const int *__restrict__ bin_start_a, const cplx<T> *__restrict__ cs_a, int ntrans, int tbegin,
    const T *__restrict__ decx, const T *__restrict__ decy, cplx<T> *__restrict__ grid_a, int nax,
    int nay, int nbx, int w, const int *__restrict__ order, int nchunk, SpreadMate mate) {
    const int *__restrict__ i0s = blockIdx.y ? mate.i0s : i0s_a;
    const T *__restrict__ kw = blockIdx.y ? static_cast<const T *>(mate.kw) : kw_a;
    const int *__restrict__ bin_start = blockIdx.y ? mate.bin_start : bin_start_a;
    const cplx<T> *__restrict__ cs = blockIdx.y ? static_cast<const cplx<T> *>(mate.cs) : cs_a;
    cplx<T> *__restrict__ grid = blockIdx.y ? static_cast<cplx<T> *>(mate.grid) : grid_a;
    static_assert(TCH % 8 == 0, "channel-group lanes need 8 | TCH");
    constexpr int CPL = TCH / 8;
    __shared__ cplx<T> s_str[SPREAD_THREADS / 64][SPREAD_CHUNK][TCH];
    __shared__ T s_kwx[SPREAD_THREADS / 64][SPREAD_CHUNK][MAX_W];
    __shared__ T s_kwy[SPREAD_THREADS / 64][SPREAD_CHUNK][SPREAD_KWP];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int og = order[blockIdx.x / nchunk];
    const int bx = (og & 0xffff) * 4 + wave, by = og >> 16;
    if (bx >= nbx) return;  // wave-uniform
    const int tbase = tbegin + (blockIdx.x % nchunk) * TCH;
    const int xl = lane & 7, cg = lane >> 3;
    const int cx = (bx << BINLOG) + xl, cy0 = by << BINLOG;
    const int *i0x = i0s, *i0y = i0s + M;
    const T *kwx = kw, *kwy = kw + M * w;
    T ar[CPL][8], ai[CPL][8];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) ar[c][k] = ai[c][k] = T(0);
    const int bxl = max((bx << BINLOG) - w + 1, 0) >> BINLOG;
    const int byl = max((by << BINLOG) - w + 1, 0) >> BINLOG;
    int rs0[3], rnc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yb = min(byl + r, by);
        const int s0 = bin_start[yb * nbx + bxl], s1 = bin_start[yb * nbx + bx + 1];
        rs0[r] = s0;
        rnc[r] = byl + r <= by ? s1 - s0 : 0;
    }
    const int nc0 = (rnc[0] + SPREAD_CHUNK - 1) / SPREAD_CHUNK, nc1 = (rnc[1] + SPREAD_CHUNK - 1) / SPREAD_CHUNK,
              nc2 = (rnc[2] + SPREAD_CHUNK - 1) / SPREAD_CHUNK;
    const int nct = nc0 + nc1 + nc2;
    auto chunk_at = [&](int c, int &n) -> int {
        int r0 = rs0[0], len = rnc[0], k = c;
        if (c >= nc0 + nc1) {
            r0 = rs0[2];
            len = rnc[2];
            k = c - nc0 - nc1;
        } else if (c >= nc0) {
            r0 = rs0[1];
            len = rnc[1];
            k = c - nc0;
        }
        n = min(SPREAD_CHUNK, len - k * SPREAD_CHUNK);
        return r0 + k * SPREAD_CHUNK;
    };
    // the padding of the wave's y-weight rows stays zero for the whole kernel (w is fixed)
    if (nct > 0)
        for (int e = lane; e < SPREAD_CHUNK * SPREAD_KWP; e += 64) (&s_kwy[wave][0][0])[e] = T(0);
    constexpr int NS = TCH / 4;                      // strengths per lane: 16 TCH / 64
    constexpr int NW = (SPREAD_CHUNK * MAX_W) / 64;  // weights per lane and dimension
    cplx<T> ps[NS];
    T pkx[NW], pky[NW];
    int pix = 0, piy = 0;
    auto request = [&](int base, int n) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int e = lane + 64 * i;
            ps[i] = {T(0), T(0)};
            if (e < n * TCH) ps[i] = cs[(int64_t)(base + e / TCH) * ntrans + tbase + e % TCH];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = lane + 64 * i;
            pkx[i] = pky[i] = T(0);
            if (e < n * w) {
                pkx[i] = kwx[(int64_t)base * w + e];
                pky[i] = kwy[(int64_t)base * w + e];
            }
        }
        if (lane < n) {
            pix = i0x[base + lane];
            piy = i0y[base + lane];
        }
    };
    int n = 0, base = 0;
    if (nct > 0) {
        base = chunk_at(0, n);
        request(base, n);
    }
    for (int c = 0; c < nct; ++c) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int e = lane + 64 * i;
            if (e < n * TCH) s_str[wave][e / TCH][e % TCH] = ps[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = lane + 64 * i;
            if (e < n * w) {
                const int j = e / w, k = e - j * w;
                s_kwx[wave][j][k] = pkx[i];
                s_kwy[wave][j][8 + k] = pky[i];
            }
        }
        const int qx = pix, qy = piy;  // lane j: footprint origin of the chunk's source j
        const int ncur = n;
        if (c + 1 < nct) {
            base = chunk_at(c + 1, n);
            request(base, n);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < ncur; ++j) {
            const int sx = __builtin_amdgcn_readlane(qx, j);
            const int oy = __builtin_amdgcn_readlane(qy, j) - cy0;  // first footprint row, block-relative
            // w > 9: the bins two back reach this block only with the far end of their footprints
            if (w > 9 && (sx + w <= (bx << BINLOG) || oy + w <= 0)) continue;  // wave-uniform
            const int dx = cx - sx;
            const T wx = (unsigned)dx < (unsigned)w ? s_kwx[wave][j][dx] : T(0);
            const T *wyp = &s_kwy[wave][j][8 - oy];  // row k of the block reads weight k - oy (or padding)
            cplx<T> cv[CPL];  // strengths times the x weight: 2 CPL multiplies instead of 8 weight products
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const cplx<T> c0 = s_str[wave][j][cg * CPL + q];
                cv[q] = {c0.re * wx, c0.im * wx};
            }
            T wy[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) wy[k] = wyp[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    ar[q][k] += cv[q].re * wy[k];
                    ai[q][k] += cv[q].im * wy[k];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the slice is rewritten by the next chunk
    }
    const T fx = decx[cx];
    const int64_t plane = (int64_t)nay * nax;
    cplx<T> *o = grid + (int64_t)(tbase + cg * CPL) * plane + (int64_t)cy0 * nax + cx;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const T f = fx * decy[cy0 + k];
#pragma unroll
        for (int q = 0; q < CPL; ++q) o[q * plane + (int64_t)k * nax] = {ar[q][k] * f, ai[q][k] * f};
    }
}

// --- 2-D spread as a small matrix product (fp64, TCH = 8 | 16, source-dense grids) ---------------------------
// Same block / bin / chunk walk and staging as k_spread2d_cg; the accumulation is
//     D[real r of the transforms][cell c] += sum_k  S[r][source k] * W[source k][cell c],     W = wx(c) wy(c),
// on the matrix pipe: v_mfma_f64_16x16x4_f64 takes 16 reals (8 complex transforms) x 16 cells x 4 sources.  Same fp64
// rate as the vector pipe -- the gain is operand traffic: the channel-group kernel reads 88 B of LDS per lane and
// source visit (8 y weights, its strengths, an x weight; LDS returns bytes per lane whether or not the address is shared)
// and is bound by exactly that at 10^6 sources; here a lane reads ONE strength real and forms ONE weight product per
// (source, cell) -- 14 B per source visit -- and the instruction broadcasts them.  Operands (guide: A[l & 15][k = l >> 4],
// B[k = l >> 4][l & 15], D[row = (l >> 4) + 4 reg][col = l & 15]): rows are mapped to (transform, re | im) so that
// a lane's four results are the complex values of transforms 2 g and 2 g + 1 (g = l >> 4) of cell l & 15 -- 16-byte
// stores, 8 lanes = one 128-byte row piece.  A block's 64 cells are four 16-cell tiles (two rows of 8) = four
// independent accumulators in flight.  Sources beyond a chunk's count enter with zero strengths.
#ifndef FV_SPREAD_MM_MINW8
#define FV_SPREAD_MM_MINW8 4  // waves per SIMD the register allocation aims at (6 / 4 spill two dozen registers and lose:
#define FV_SPREAD_MM_MINW16 3 // 0.55 against 0.37 ms per launch at 10^6 sources)
#endif
template <int TCH>
__global__ __launch_bounds__(SPREAD_THREADS, TCH == 8 ? FV_SPREAD_MM_MINW8 : FV_SPREAD_MM_MINW16) void k_spread2d_mm(
    int64_t M, const int *__restrict__ i0s_a, const double *__restrict__ kw_a,
    const int *__restrict__ bin_start_a, const cplx<double> *__restrict__ cs_a, int ntrans, int tbegin,
    const double *__restrict__ decx, const double *__restrict__ decy, cplx<double> *__restrict__ grid_a, int nax,
    int nay, int nbx, int w, const int *__restrict__ order, int nchunk, SpreadMate mate) {
    using T = double;
    using d4 = double __attribute__((ext_vector_type(4)));
    const int *__restrict__ i0s = blockIdx.y ? mate.i0s : i0s_a;
    const T *__restrict__ kw = blockIdx.y ? static_cast<const T *>(mate.kw) : kw_a;
    const int *__restrict__ bin_start = blockIdx.y ? mate.bin_start : bin_start_a;
    const cplx<T> *__restrict__ cs = blockIdx.y ? static_cast<const cplx<T> *>(mate.cs) : cs_a;
    cplx<T> *__restrict__ grid = blockIdx.y ? static_cast<cplx<T> *>(mate.grid) : grid_a;
    static_assert(TCH == 8 || TCH == 16, "8 transforms per 16-row operand tile");
    constexpr int MT = TCH / 8;  // operand tiles of 16 reals
    __shared__ T s_sr[SPREAD_THREADS / 64][SPREAD_CHUNK][MT * 16];
    __shared__ T s_kwx[SPREAD_THREADS / 64][SPREAD_CHUNK][MAX_W];
    __shared__ T s_kwy[SPREAD_THREADS / 64][SPREAD_CHUNK][MAX_W];
    __shared__ int s_i0[SPREAD_THREADS / 64][SPREAD_CHUNK][2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int og = order[blockIdx.x / nchunk];
    const int bx = (og & 0xffff) * 4 + wave, by = og >> 16;
    if (bx >= nbx) return;  // wave-uniform
    const int tbase = tbegin + (blockIdx.x % nchunk) * TCH;
    const int g = lane >> 4;                                           // operand k index / result row group
    const int cx = (bx << BINLOG) + (lane & 7), cy0 = (by << BINLOG) + ((lane >> 3) & 1);  // cell of tile t: (cx, cy0 + 2 t)
    const int *i0x = i0s, *i0y = i0s + M;
    const T *kwx = kw, *kwy = kw + M * w;
    d4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = d4{0.0, 0.0, 0.0, 0.0};
    const int bxl = max((bx << BINLOG) - w + 1, 0) >> BINLOG;
    const int byl = max((by << BINLOG) - w + 1, 0) >> BINLOG;
    int rs0[3], rnc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yb = min(byl + r, by);
        const int s0 = bin_start[yb * nbx + bxl], s1 = bin_start[yb * nbx + bx + 1];
        rs0[r] = s0;
        rnc[r] = byl + r <= by ? s1 - s0 : 0;
    }
    // The sources of the <= 3 bin rows form ONE visit list (rows one after the other) cut into chunks of 16: at a few
    // sources per bin three separate lists made two or three half-empty chunks per block, and the chunks are a serial
    // chain of load round trips (the kernel is bound by that latency at 10^6 sources, not by its arithmetic).
    const int L0 = rnc[0], L01 = rnc[0] + rnc[1], ntot = L01 + rnc[2];
    const int nct = (ntot + SPREAD_CHUNK - 1) / SPREAD_CHUNK;
    auto src_of = [&](int i) -> int {  // visit-list element i -> sorted source index
        return i < L0 ? rs0[0] + i : i < L01 ? rs0[1] + (i - L0) : rs0[2] + (i - L01);
    };
    // weights and origins of slots a short chunk leaves untouched must be finite and in range from the start
    // (their strengths are zero; 0 x NaN would not be)
    if (nct > 0) {
        for (int e = lane; e < SPREAD_CHUNK * MAX_W; e += 64) {
            (&s_kwx[wave][0][0])[e] = T(0);
            (&s_kwy[wave][0][0])[e] = T(0);
        }
        if (lane < SPREAD_CHUNK * 2) (&s_i0[wave][0][0])[lane] = 0;
    }
    constexpr int NS = TCH / 4;                      // strengths per lane: 16 TCH / 64
    constexpr int NW = (SPREAD_CHUNK * MAX_W) / 64;  // weights per lane and dimension
    cplx<T> ps[NS];
    T pkx[NW], pky[NW];
    int pix = 0, piy = 0;
    int wj[NW], wk[NW];  // weight element e = lane + 64 i of a chunk: slot e / w, tap e % w (the same for every chunk)
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int e = lane + 64 * i;
        wj[i] = e / w;
        wk[i] = e - wj[i] * w;
    }
    auto request = [&](int base, int n) {  // base: first visit-list element of the chunk
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int e = lane + 64 * i;
            ps[i] = {T(0), T(0)};
            if (e < n * TCH) ps[i] = cs[(int64_t)src_of(base + e / TCH) * ntrans + tbase + e % TCH];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            pkx[i] = pky[i] = T(0);
            if (wj[i] < n) {
                const int64_t at = (int64_t)src_of(base + wj[i]) * w + wk[i];
                pkx[i] = kwx[at];
                pky[i] = kwy[at];
            }
        }
        if (lane < n) {
            const int sidx = src_of(base + lane);
            pix = i0x[sidx];
            piy = i0y[sidx];
        }
    };
    int n = 0, base = 0;
    if (nct > 0) {
        n = min(SPREAD_CHUNK, ntot);
        request(0, n);
    }
    for (int c = 0; c < nct; ++c) {
        // ---- registers -> LDS: every slot's strengths (zeros beyond the chunk's count), this chunk's weights ------
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int e = lane + 64 * i;  // (source j, transform tq) of the chunk, zero beyond n
            const int j = e / TCH, tq = e % TCH;
            // operand row of (transform, part): tile tq / 8, row (tq' / 2) + 4 (2 (tq' % 2) + part), tq' = tq % 8
            const int row = (tq >> 3) * 16 + ((tq & 7) >> 1) + 8 * (tq & 1);
            s_sr[wave][j][row] = ps[i].re;
            s_sr[wave][j][row + 4] = ps[i].im;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (wj[i] < n) {
                s_kwx[wave][wj[i]][wk[i]] = pkx[i];
                s_kwy[wave][wj[i]][wk[i]] = pky[i];
            }
        }
        if (lane < n) {
            s_i0[wave][lane][0] = pix;
            s_i0[wave][lane][1] = piy;
        }
        const int ncur = n;
        if (c + 1 < nct) {
            base = (c + 1) * SPREAD_CHUNK;
            n = min(SPREAD_CHUNK, ntot - base);
            request(base, n);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- four sources per instruction ------------------------------------------------------------------
        const int nq = (ncur + 3) >> 2;
        for (int q = 0; q < nq; ++q) {
            const int sj = 4 * q + g;  // this lane's source of the k dimension (slots beyond ncur: zero strengths)
            const int dx = cx - s_i0[wave][sj][0], dy0 = cy0 - s_i0[wave][sj][1];
            const T wx = (unsigned)dx < (unsigned)w ? s_kwx[wave][sj][dx] : T(0);
            T a[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = s_sr[wave][sj][m * 16 + (lane & 15)];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int dy = dy0 + 2 * t;
                const T b = (unsigned)dy < (unsigned)w ? wx * s_kwy[wave][sj][dy] : T(0);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b, acc[m][t], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the slice is rewritten by the next chunk
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // results: lane (g, cell) holds transforms 2 g, 2 g + 1 of every operand tile as (re, im) pairs
    const T fx = decx[cx];
    const int64_t plane = (int64_t)nay * nax;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int cy = cy0 + 2 * t;
        const T f = fx * decy[cy];
        cplx<T> *o = grid + (int64_t)tbase * plane + (int64_t)cy * nax + cx;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            o[(int64_t)(m * 8 + 2 * g) * plane] = {acc[m][t][0] * f, acc[m][t][1] * f};
            o[(int64_t)(m * 8 + 2 * g + 1) * plane] = {acc[m][t][2] * f, acc[m][t][3] * f};
        }
    }
}

// --- 3-D spread (gather): as k_spread2d, one wave per 8x8 (x, y) block of ONE z-plane of A ------
// grid (ceil(nbx / 4), nby, na_z * chunks); bin index = (bz * nby + by) * nbx + bx.  The z weight of
// a staged source is folded into its strengths while staging (it is wave-uniform: one z-plane).
template <typename T, int TCH>
__global__ __launch_bounds__(SPREAD_THREADS) void k_spread3d(
    int64_t M, const int *__restrict__ i0s, const T *__restrict__ kw,
    const int *__restrict__ bin_start, const cplx<T> *__restrict__ cs, int ntrans, int tbegin,
    int nchunk, const T *__restrict__ decx, const T *__restrict__ decy,
    const T *__restrict__ decz, cplx<T> *__restrict__ grid, int nax, int nay, int naz, int nbx,
    int nby, int w, const int *__restrict__ row_ext) {
    __shared__ cplx<T> s_str[SPREAD_THREADS / 64][SPREAD_CHUNK][TCH];
    __shared__ T s_kw[SPREAD_THREADS / 64][SPREAD_CHUNK][2][MAX_W];
    __shared__ int s_i0[SPREAD_THREADS / 64][SPREAD_CHUNK][2];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int bx = blockIdx.x * 4 + wave, by = blockIdx.y;
    if (bx >= nbx) return;  // wave-uniform
    // source disc (Nufft3::build_block_order): blocks outside the block row's extent are never touched by a source and
    // never read by the x-pass -- not written either (whole 4-block groups, as in the 2-D launch list)
    if (row_ext && (32 * (int)blockIdx.x >= row_ext[2 * by + 1] || 32 * (int)blockIdx.x + 32 <= row_ext[2 * by])) return;
    const int cz = blockIdx.z / nchunk;
    const int tbase = tbegin + (blockIdx.z % nchunk) * TCH;
    const int cx = (bx << BINLOG) + (lane & 7), cy = (by << BINLOG) + (lane >> 3);
    const int *i0x = i0s, *i0y = i0s + M, *i0z = i0s + 2 * M;
    const T *kwx = kw, *kwy = kw + M * w, *kwz = kw + 2 * M * w;
    T ar[TCH], ai[TCH];
#pragma unroll
    for (int q = 0; q < TCH; ++q) ar[q] = ai[q] = T(0);
    const int bxl = max((bx << BINLOG) - w + 1, 0) >> BINLOG;
    const int byl = max((by << BINLOG) - w + 1, 0) >> BINLOG;
    const int bzl = max(cz - w + 1, 0) >> BINLOG, bzh = cz >> BINLOG;
    for (int zb = bzl; zb <= bzh; ++zb) {
        for (int yb = byl; yb <= by; ++yb) {
            const int rowb = (zb * nby + yb) * nbx;
            const int s0 = bin_start[rowb + bxl], s1 = bin_start[rowb + bx + 1];
            for (int base = s0; base < s1; base += SPREAD_CHUNK) {
                const int n = min(SPREAD_CHUNK, s1 - base);
                for (int e = lane; e < n * TCH; e += 64) {
                    const int j = e / TCH, q = e % TCH;
                    const int dz = cz - i0z[base + j];
                    const T kz = (unsigned)dz < (unsigned)w ? kwz[(int64_t)(base + j) * w + dz] : T(0);
                    const cplx<T> c = cs[(int64_t)(base + j) * ntrans + tbase + q];
                    s_str[wave][j][q] = {c.re * kz, c.im * kz};
                }
                for (int e = lane; e < n * w; e += 64) {
                    const int j = e / w, k = e - j * w;
                    s_kw[wave][j][0][k] = kwx[(int64_t)(base + j) * w + k];
                    s_kw[wave][j][1][k] = kwy[(int64_t)(base + j) * w + k];
                }
                if (lane < n) {
                    s_i0[wave][lane][0] = i0x[base + lane];
                    s_i0[wave][lane][1] = i0y[base + lane];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int j = 0; j < n; ++j) {
                    const int dx = cx - s_i0[wave][j][0], dy = cy - s_i0[wave][j][1];
                    T wt = T(0);
                    if ((unsigned)dx < (unsigned)w && (unsigned)dy < (unsigned)w)
                        wt = s_kw[wave][j][0][dx] * s_kw[wave][j][1][dy];
#pragma unroll
                    for (int q = 0; q < TCH; ++q) {
                        const cplx<T> cv = s_str[wave][j][q];
                        ar[q] += cv.re * wt;
                        ai[q] += cv.im * wt;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    const T f = decx[cx] * decy[cy] * (decz ? decz[cz] : T(1));  // (no inner kernel along z when the targets sum the planes directly)
    const int64_t plane = (int64_t)naz * nay * nax;
    cplx<T> *o = grid + (int64_t)tbase * plane + ((int64_t)cz * nay + cy) * nax + cx;
#pragma unroll
    for (int q = 0; q < TCH; ++q) o[q * plane] = {ar[q] * f, ai[q] * f};
}

// --- pruned row FFT -----------------------------------------------------------------------------
// out[row][j] = sum_{ia < n_in} in[row][ia] exp(+2 pi i (ia - n_in/2)(j - n_out/2) / n2),
// n2 = P * Q, Q = 2^logQ: sub-FFTs of length Q live in LDS (rows padded against bank conflicts)
// and are computed by in-place DIF passes of radix 16/8 held in registers (result digit-reversed).
// LDS padding: one element per 16 and one more per 256, so that the unit-stride last pass, the
// stride-16 middle pass and the stride-256 digit-reversed read-out all spread over the banks.
__host__ __device__ inline int fft_pidx(int i) { return i + (i >> 4) + (i >> 8); }

// Complex arithmetic of the FFT passes.  fp64: plain expressions.  fp32 (FV_PK_F32, default on): one complex value =
// one even-aligned VGPR pair and every operation a packed instruction -- v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with
// the op_sel / neg modifiers doing the swizzles of a complex product (a.re w, then +- a.im (w.im, w.re)) and of a
// multiplication by i for free.  Scalar fp32 issues at the fp64 rate on gfx950 (16 lanes per clock; the 157 TF vector
// peak is the packed rate), so the fp32 passes cost what the fp64 ones do unless they are packed; the compiler packs the
// additions by itself but moves the halves around for every product (490 v_mov per thread when tried), hence the
// inline assembly for the products.  A radix-16 butterfly: 84 packed instructions instead of 168 scalar ones.
#ifndef FV_PK_F32
#define FV_PK_F32 1
#endif
using f2v = float __attribute__((ext_vector_type(2)));
template <typename T>
struct PkF32 {
    static constexpr bool on = false;
};
template <>
struct PkF32<float> {
    static constexpr bool on = FV_PK_F32 != 0;
};
__device__ inline cplx<float> pk_add(cplx<float> a, cplx<float> b) {
    return __builtin_bit_cast(cplx<float>, __builtin_bit_cast(f2v, a) + __builtin_bit_cast(f2v, b));
}
__device__ inline cplx<float> pk_sub(cplx<float> a, cplx<float> b) {
    return __builtin_bit_cast(cplx<float>, __builtin_bit_cast(f2v, a) - __builtin_bit_cast(f2v, b));
}
__device__ inline cplx<float> pk_isub(cplx<float> a, cplx<float> b) {  // i (a - b)
    f2v r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,1]"
        : "=v"(r)
        : "v"(__builtin_bit_cast(f2v, a)), "v"(__builtin_bit_cast(f2v, b)));
    return __builtin_bit_cast(cplx<float>, r);
}
// (Both instructions of a product sit in ONE asm statement: between two statements the compiler's hazard recogniser, which
// cannot see inside, puts an s_nop -- 97 of them per thread in the folded x-pass, an issue slot each, when they were two.)
__device__ inline cplx<float> pk_mul(cplx<float> a, cplx<float> w) {  // a w
    f2v t, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(t), "=v"(r)
        : "v"(__builtin_bit_cast(f2v, a)), "v"(__builtin_bit_cast(f2v, w)));
    return __builtin_bit_cast(cplx<float>, r);
}
__device__ inline cplx<float> pk_mul_s(cplx<float> a, cplx<float> w) {  // a w, w wave-uniform (scalar registers)
    f2v t, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(t), "=v"(r)
        : "v"(__builtin_bit_cast(f2v, a)), "s"(__builtin_bit_cast(f2v, w)));
    return __builtin_bit_cast(cplx<float>, r);
}
__device__ inline cplx<float> pk_mac(cplx<float> acc, cplx<float> a, cplx<float> w) {  // acc + a w
    f2v t, r;
    asm("v_pk_fma_f32 %0, %2, %3, %4 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(t), "=v"(r)
        : "v"(__builtin_bit_cast(f2v, a)), "v"(__builtin_bit_cast(f2v, w)), "v"(__builtin_bit_cast(f2v, acc)));
    return __builtin_bit_cast(cplx<float>, r);
}
__device__ inline cplx<float> pk_mac_s(cplx<float> acc, cplx<float> a, cplx<float> w) {  // acc + a w, w wave-uniform
    f2v t, r;
    asm("v_pk_fma_f32 %0, %2, %3, %4 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(t), "=v"(r)
        : "v"(__builtin_bit_cast(f2v, a)), "s"(__builtin_bit_cast(f2v, w)), "v"(__builtin_bit_cast(f2v, acc)));
    return __builtin_bit_cast(cplx<float>, r);
}
// the passes' products and multiply-adds: packed in fp32, plain otherwise (_u: the factor is wave-uniform -- it stays in
// scalar registers instead of being copied into a vector pair)
template <typename T>
__device__ inline cplx<T> xmul(cplx<T> a, cplx<T> w) {
    if constexpr (PkF32<T>::on)
        return pk_mul(a, w);
    else
        return cmul(a, w);
}
template <typename T>
__device__ inline cplx<T> xmul_u(cplx<T> a, cplx<T> w) {
    if constexpr (PkF32<T>::on)
        return pk_mul_s(a, w);
    else
        return cmul(a, w);
}
template <typename T>
__device__ inline cplx<T> xmac(cplx<T> acc, cplx<T> a, cplx<T> w) {
    if constexpr (PkF32<T>::on)
        return pk_mac(acc, a, w);
    else
        return {acc.re + (a.re * w.re - a.im * w.im), acc.im + (a.re * w.im + a.im * w.re)};
}

template <typename T>
__device__ inline cplx<T> cmul_root16(cplx<T> d, int k) {  // d * exp(+2 pi i k / 16), 0 <= k < 8
    constexpr T C1 = T(0.92387953251128675613), S1 = T(0.38268343236508977173),
                H = T(0.70710678118654752440);
    switch (k) {
        case 0: return d;
        case 1: return {d.re * C1 - d.im * S1, d.re * S1 + d.im * C1};
        case 2: return {(d.re - d.im) * H, (d.re + d.im) * H};
        case 3: return {d.re * S1 - d.im * C1, d.re * C1 + d.im * S1};
        case 4: return {-d.im, d.re};
        case 5: return {-d.re * S1 - d.im * C1, d.re * C1 - d.im * S1};
        case 6: return {(-d.re - d.im) * H, (d.re - d.im) * H};
        default: return {-d.re * C1 - d.im * S1, d.re * S1 - d.im * C1};
    }
}

// In-register radix-2 DIF network of size R (inverse sign); X[k] ends up in v[bitrev_R(k)].
template <typename T, int R>
__device__ inline void dif_regs(cplx<T> *v) {
    if constexpr (R > 1) {
        constexpr int H = R / 2;
#pragma unroll
        for (int k = 0; k < H; ++k) {
            const cplx<T> a = v[k], b = v[k + H];
            if constexpr (PkF32<T>::on) {
                constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, HH = 0.70710678118654752440f;
                const int kk = k * (16 / R);  // compile-time after unrolling: d exp(+2 pi i kk / 16), d = a - b
                v[k] = pk_add(a, b);
                if (kk == 0)
                    v[k + H] = pk_sub(a, b);
                else if (kk == 4)
                    v[k + H] = pk_isub(a, b);  // the rotation rides on the subtraction
                else {
                    const cplx<float> wr = kk == 1   ? cplx<float>{C1, S1}
                                           : kk == 2 ? cplx<float>{HH, HH}
                                           : kk == 3 ? cplx<float>{S1, C1}
                                           : kk == 5 ? cplx<float>{-S1, C1}
                                           : kk == 6 ? cplx<float>{-HH, HH}
                                                     : cplx<float>{-C1, S1};
                    v[k + H] = pk_mul_s(pk_sub(a, b), wr);
                }
            } else {
                v[k] = {a.re + b.re, a.im + b.im};
                v[k + H] = cmul_root16<T>(cplx<T>{a.re - b.re, a.im - b.im}, k * (16 / R));
            }
        }
        dif_regs<T, H>(v);
        dif_regs<T, H>(v + H);
    }
}

constexpr int bitrev_small(int k, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((k >> i) & 1) << (bits - 1 - i);
    return r;
}

// One radix-R butterfly group of a DIF pass with current span L = R << logLR on LDS row rb.
template <typename T, int LOGR>
__device__ inline void fft_pass_item(cplx<T> *rb, int u, int logLR, const cplx<T> *__restrict__ tw,
                                     int twmul) {
    constexpr int R = 1 << LOGR;
    const int stride = 1 << logLR;
    const int g = u >> logLR, j = u & (stride - 1);
    const int base = (g << (logLR + LOGR)) + j;
    const cplx<T> w = tw[j * twmul];  // issued first: its latency hides under the LDS reads
    cplx<T> v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = rb[fft_pidx(base + k * stride)];
    dif_regs<T, R>(v);
    if (logLR == 0) {
#pragma unroll
        for (int k = 0; k < R; ++k) rb[fft_pidx(base + k)] = v[bitrev_small(k, LOGR)];
    } else {
        cplx<T> wk = w;
        rb[fft_pidx(base)] = v[0];
#pragma unroll
        for (int k = 1; k < R; ++k) {
            rb[fft_pidx(base + k * stride)] = cmul(v[bitrev_small(k, LOGR)], wk);
            wk = cmul(wk, w);
        }
    }
}

// position of frequency k after the in-place DIF passes (mixed-radix digit reversal)
template <typename Args>
__device__ inline int fft_digit_pos(int k, const Args &a) {
    int pos = 0, span = a.logQ;
    for (int s = 0; s < a.npass; ++s) {
        const int rl = a.radix_log[s];
        span -= rl;
        pos += (k & ((1 << rl) - 1)) << span;
        k >>= rl;
    }
    return pos;
}

// --- pruned row FFT, decimation in frequency over the P residues -------------------------------
// Output l = P k' + p needs only  Y_p = FFT_Q(fold_Q(x[ia] w^{ia p})):
//     X[l] = w^{-(n_in/2) l} Y_p[k' mod Q],      w = exp(2 pi i / n2),
// so every (row group, residue p) is an independent, accumulator-free job: load (twiddle + fold;
// a plain copy for p = 0 and n_in <= Q), DIF passes, strided write of its own outputs.  108 VGPRs,
// Q/8 threads per row (up to 512): 8 elements of LDS per thread keeps 4 waves per SIMD.  The P jobs of
// a row group get block ids that differ by 8, i.e. the same XCD: their interleaved 16-B stores
// meet in that XCD's L2 before they are written back.
struct RowDifArgs {
    int n_in, n_out, n2, P, Q, logQ, tpr, rpw;
    int npass, radix_log[4];
    int lds_row, colmode;
    int cnt;  // outputs per residue: row position of l is ((l + n_out/2) mod P) cnt + (l + n_out/2) / P;
              // 0 = natural order (position l + n_out/2)
    int64_t nrows, rpp, rpp_valid, in_plane, in_row, in_elem, out_pitch;  // rows k >= rpp_valid of a plane are padding
    // Column-blocked planes (log2 of the block width in elements, 0 = plain row-major [row][pitch]): element (row k,
    // position x) of a plane sits at (x >> b) (rows << b) + (k << b) + (x & (2^b - 1)) -- 64-byte pieces of a row, and
    // the pieces of one block of columns contiguous over the rows.  The x-pass writes it (out_blk), the column-mode
    // y-pass reads it (in_blk): its 4 columns are then ONE contiguous run instead of a 64-B piece per 80-KB row.
    int in_blk, out_blk;
    int jobs_per_xcd;  // row mode: rows per XCD (a multiple of the rows per workgroup), see k_rowfft_st
    const void *in1;  // gang launch (grid.y = 2): input / output of the second, identically shaped problem
    void *out1;
    // Column plan (row mode, blocked output; see Nufft3::arm_columns): ctab[(plane / ctab_tpol) ctab_stride + position]
    // = 1 + the ELEMENT INDEX, inside a row's blocked output, of the compact column an output position is stored at
    // ((c >> b) (rows << b) + (c & (2^b - 1)) for compact column c), 0 = no target's footprint reads that column: not stored.
    const int *ctab;
    int ctab_stride, ctab_tpol;
    // Row extents (row mode, first pass of a 2-D transform whose sources lie in a disc; Nufft3::disc_radius): the input
    // row k of a plane is non-zero -- and was written by the spread -- only in [row_ext[2 (k >> 3)], row_ext[2 (k >> 3) + 1]).
    const int *row_ext;
    // Output mask (column mode under a column plan): bit c of omask[(((plane / omask_tpol) omask_nblk + column block) P +
    // residue) words + c / 64] says whether ANY target reads one of the block's outputs k' in [16 c, 16 c + 16) of that
    // residue; the stores of the other chunks -- most of them: footprints are w cells high -- are branched over.
    const unsigned long long *omask;
    int omask_tpol, omask_nblk;
};

template <typename T>
__global__ __launch_bounds__(512, 4) void k_rowfft_dif(const cplx<T> *__restrict__ in,
                                                     cplx<T> *__restrict__ out,
                                                     const cplx<T> *__restrict__ tw, RowDifArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fft_smem[];
    cplx<T> *smem = reinterpret_cast<cplx<T> *>(fft_smem);
    const int tid = threadIdx.x;
    const int vb = blockIdx.x;
    const int tt = vb >> 3;
    const int p = tt % a.P;
    const int64_t grp = (int64_t)(tt / a.P) * 8 + (vb & 7);
    const int64_t row0 = grp * a.rpw;
    if (row0 >= a.nrows) return;  // workgroup-uniform
    const int Q = a.Q, n2 = a.n2;
    const int r = tid / a.tpr, lane = tid % a.tpr;

    // ---- load: rb[q] = sum_k x[q + k Q] w^{(q + k Q) p} -------------------------------------------
    {
        const int rr = a.colmode ? (tid & (a.rpw - 1)) : r;  // colmode: lanes over adjacent columns
        const int q0 = a.colmode ? tid / a.rpw : lane;
        const int64_t rw = row0 + rr;
        const bool ok = rw < a.nrows && rw % a.rpp < a.rpp_valid;
        const cplx<T> *rin = in + (ok ? (rw / a.rpp) * a.in_plane + (rw % a.rpp) * a.in_row : 0);
        cplx<T> *crb = smem + (int64_t)rr * a.lds_row;
        for (int q = q0; q < Q; q += a.tpr) {
            cplx<T> v = {T(0), T(0)};
            if (ok && q < a.n_in) {
                v = rin[(int64_t)q * a.in_elem];
                if (p) v = cmul(v, tw[q * p]);  // q p < Q P = n2
                int rk = p;  // (k p) mod P
                for (int ia = q + Q; ia < a.n_in; ia += Q) {
                    int ti = q * p + Q * rk;
                    if (ti >= n2) ti -= n2;
                    const cplx<T> x = cmul(rin[(int64_t)ia * a.in_elem], tw[ti]);
                    v = {v.re + x.re, v.im + x.im};
                    rk += p;
                    if (rk >= a.P) rk -= a.P;
                }
            }
            crb[fft_pidx(q)] = v;
        }
    }
    __syncthreads();

    cplx<T> *rb = smem + (int64_t)r * a.lds_row;
    int logL = a.logQ;
    for (int s = 0; s < a.npass; ++s) {
        const int rl = a.radix_log[s];
        const int items = Q >> rl, logLR = logL - rl;
        const int twmul = a.P << (a.logQ - logL);  // n2 / L
        if (rl == 4) {
            for (int u = lane; u < items; u += a.tpr) fft_pass_item<T, 4>(rb, u, logLR, tw, twmul);
        } else if (rl == 3) {
            for (int u = lane; u < items; u += a.tpr) fft_pass_item<T, 3>(rb, u, logLR, tw, twmul);
        } else if (rl == 2) {
            for (int u = lane; u < items; u += a.tpr) fft_pass_item<T, 2>(rb, u, logLR, tw, twmul);
        } else {
            for (int u = lane; u < items; u += a.tpr) fft_pass_item<T, 1>(rb, u, logLR, tw, twmul);
        }
        __syncthreads();
        logL = logLR;
    }

    // ---- this residue's outputs: l = l0 + P i, l in [-n_out/2, n_out/2) -----------------------------
    const int64_t row = row0 + r;
    if (row >= a.nrows || row % a.rpp >= a.rpp_valid) return;
    const int half_n = a.n_out / 2;
    int d0 = (p + half_n) % a.P;  // (p - (-half_n)) mod P
    const int lfirst = -half_n + d0 + a.P * lane;
    if (lfirst >= a.n_out - half_n) return;
    const int hshift = a.n_in / 2;
    int ti = (int)((-(int64_t)hshift * lfirst) % n2);
    if (ti < 0) ti += n2;
    cplx<T> t = tw[ti];
    int si = (int)((-(int64_t)hshift * a.P * a.tpr) % n2);
    if (si < 0) si += n2;
    const cplx<T> step = tw[si];
    // residue-major storage: l = P kq + p sits at ((p + half_n) mod P) cnt + (p + half_n) / P + kq
    cplx<T> *rout = out + ((row / a.rpp) * a.rpp_valid + row % a.rpp) * a.out_pitch;
    rout += a.cnt ? ((p + half_n) % a.P) * a.cnt + (p + half_n) / a.P : p + half_n;
    const int ostep = a.cnt ? 1 : a.P;  // position of l = P kq + p: base + kq (residue-major) or + P kq
    // k' = (l - p) / P advances by tpr per step: no division inside the loop
    int kq = (lfirst - p) / a.P;
    for (int l = lfirst; l < a.n_out - half_n; l += a.P * a.tpr, kq += a.tpr) {
        rout[kq * ostep] = cmul(rb[fft_pidx(fft_digit_pos(kq & (Q - 1), a))], t);
        t = cmul(t, step);
    }
}

// --- pruned row FFT, register-resident (Q = 512 .. 4096) ---------------------------------------
// Same job decomposition as k_rowfft_dif (one workgroup-row per (row, residue p)), but the three
// radix passes keep their operands in registers and LDS is only the exchange between passes:
//   pass 1  item u = n' in [0, Q/R1):  x[u + n1 Q/R1] straight from global memory (coalesced over u,
//           zero beyond n_in, twiddle/fold for p > 0), radix-R1 butterfly, twiddle w_Q^{u k1};
//   pass 2  item (k1, j3): radix R2 over the sub-sequence of length Q/R1, twiddle;
//   pass 3  item v = k1 + R1 k2: radix R3, then X[v + k3 Q/R3] goes straight to global memory
//           (coalesced over v; only the wanted |l| range is written).
// The exchange moves real and imaginary parts one after the other through one T-typed buffer of
// R1*A slots per row (8.4 KiB at Q = 1024 fp64 instead of 17.5 KiB for a complex row), so twice as
// many rows are in flight per CU, and every element crosses LDS 4 times instead of 8.
// Slot of element (k1, j2, j3) of row r = r ROW + k1 A + j2 B + j3: A, B, ROW searched with a bank
// model of ds_read_b64 (2 x 32 lanes, 32 slot classes) and ds_write_b64 (4 x 16 lanes, 16 classes)
// so that the pass-1 writes and the pass-2 accesses are conflict-free and the pass-3 reads at
// most 2-way, in row mode and in column mode (512 threads, lanes interleave 8 rows, so that a
// workgroup reads whole 128-B lines of 8 adjacent columns).
template <int LOGQ, bool COL>
struct StPlan;
template <bool COL>
struct StPlan<9, COL> {  // column mode interleaves 8 rows over the lanes: its own padding
    static constexpr int R1 = 8, R2 = 8, R3 = 8, TPR = 64, A = COL ? 72 : 76, B = COL ? 8 : 9,
                         ROW = COL ? 577 : 608;
};
template <bool COL>
struct StPlan<10, COL> {
    static constexpr int R1 = 16, R2 = 8, R3 = 8, TPR = 64, A = COL ? 72 : 66, B = 8, ROW = COL ? 1153 : 1056;
};
template <bool COL>
struct StPlan<11, COL> {
    static constexpr int R1 = 16, R2 = 16, R3 = 8, TPR = 128, A = COL ? 132 : 130, B = 8, ROW = COL ? 2115 : 2080;
};
template <bool COL>
struct StPlan<12, COL> {
    static constexpr int R1 = 16, R2 = 16, R3 = 16, TPR = 256, A = 258, B = 16, ROW = 4128;
};
// FV_FFT_STAMPS (diagnostic builds only): every wave of k_rowfft_st leaves s_memrealtime stamps (100 MHz) at its phase
// boundaries in a device array that fv_debug_stamps() copies out -- where a job's time goes (load, three radix passes,
// two exchanges, stores).  Never defined in the product build.
#ifdef FV_FFT_STAMPS
__device__ unsigned long long fv_stamps[(size_t)10 << 18];  // 7 realtime stamps, tag, realtime at the wave's first instruction / after its stores drained
__device__ unsigned int fv_stamp_count;
#define FV_STAMP(slot)                                                                          \
    do {                                                                                         \
        if ((threadIdx.x & 63) == 0 && stamp_idx < (1u << 18)) {                                 \
            fv_stamps[(size_t)stamp_idx * 10 + (slot)] = __builtin_amdgcn_s_memrealtime();         \
            if ((slot) == 0) fv_stamps[(size_t)stamp_idx * 10 + 8] = stamp_entry;                  \
            if ((slot) == 6) {                                                                   \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 \
                fv_stamps[(size_t)stamp_idx * 10 + 9] = __builtin_amdgcn_s_memrealtime();         \
            }                                                                                    \
        }                                                                                        \
    } while (0)
#else
#define FV_STAMP(slot)
#endif
#ifndef FV_PAIR_DEFAULT
#define FV_PAIR_DEFAULT 0  // paired residue jobs (k_rowfft_st, PAIR): off since round 4 -- with the pass-2 table and two lanes the
                           // unpaired jobs run C3 3 % faster (843 -> 816, 863 -> 837, 869 -> 841 ms per step), C4 2 %, C5 and the
                           // scattered array 1 %; FFTVIS_HIP_PAIR = 1 (row mode) | 2 (column mode too) turn it on
#endif
#ifndef FV_ST_FOLD_SWEEPS3
#define FV_ST_FOLD_SWEEPS3 1  // three sweeps of a fold chunk in one round trip where it has exactly three (k_rowfft_st, FOLD)
#endif
#ifndef FV_ST_TW2_LDS
#define FV_ST_TW2_LDS 1  // pass-2 twiddles from a small LDS table (k_rowfft_st, TW2)
#endif
#ifndef FV_ST_MAP2
#define FV_ST_MAP2 1  // pass-2 items assigned so that 16 contiguous lanes use 16 bank classes (k_rowfft_st, MAP2)
#endif
#ifndef FV_ST_CX32
#define FV_ST_CX32 0  // fp32: whole complex values through the LDS exchange (k_rowfft_st, CX): measured slower, see there
#endif
#ifndef FV_ST_MINW12
#define FV_ST_MINW12 3  // waves per SIMD targeted by the register allocation of the Q = 4096 kernels
#endif
constexpr int ST_THREADS = 256;      // row mode
constexpr int ST_THREADS_COL = 512;  // column mode
#ifndef FV_COL11_THREADS
#define FV_COL11_THREADS 512
#endif
constexpr int st_threads(int logq, bool col, bool pair = false) {
    // PAIR (two residues per job, see k_rowfft_st): a column-mode workgroup keeps its columns and doubles its threads
    return (!col ? ST_THREADS : logq == 12 ? 1024 : logq == 11 ? FV_COL11_THREADS : ST_THREADS_COL) * (col && pair ? 2 : 1);
}
constexpr int ilog2_c(int v) { return v <= 1 ? 0 : 1 + ilog2_c(v / 2); }

// A pointer every lane of the wave holds the same value of, moved to scalar registers: global loads /
// stores through it take the scalar-base + 32-bit-offset form (one VALU op per address instead of a
// 64-bit multiply-add chain, and no address pairs kept in vector registers).
template <typename P>
__device__ inline P *wave_uniform_ptr(P *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<P *>(((uint64_t)hi << 32) | lo);
}

// Buffer-descriptor access to one row of complex elements (row mode: the row belongs to whole waves, so base
// and extent are wave-uniform).  The hardware range check replaces the clamp / compare / select per element:
// a load at an element index outside [0, n) -- negative indices wrap to huge unsigned offsets -- returns
// zero, a store there is dropped.  n = 0 switches the row off altogether.
template <typename T>
struct RowBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    // n_elems < 0: no extent (column mode: lanes address several rows from one base and mask by index -1)
    __device__ RowBuf(const cplx<T> *base, int64_t n_elems) {
        const cplx<T> *b = wave_uniform_ptr(base);
        const uint32_t bytes = __builtin_amdgcn_readfirstlane(n_elems < 0 ? 0xFFFFFFF0u : (uint32_t)(n_elems * (int64_t)sizeof(cplx<T>)));
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx<T> *>(b), 0, bytes, 0x00020000);
    }
    __device__ cplx<T> load(int idx) const {
#if defined(FV_ABL) && (FV_ABL & 2)  // diagnostic build: every load reads element 0 (cached): what the input traffic costs
        const uint32_t off = idx < 0 ? 0xfffffff0u : 0u;
#else
        const uint32_t off = (uint32_t)idx * (uint32_t)sizeof(cplx<T>);
#endif
        if constexpr (sizeof(T) == 8) {
            using v4 = unsigned int __attribute__((ext_vector_type(4)));
            const v4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            return __builtin_bit_cast(cplx<T>, r);
        } else {
            using v2 = unsigned int __attribute__((ext_vector_type(2)));
            const v2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
            return __builtin_bit_cast(cplx<T>, r);
        }
    }
    __device__ void store(int idx, cplx<T> v) const {
#if defined(FV_ABL) && (FV_ABL & 1)  // diagnostic build: outputs are computed, kept alive, and dropped by the range check
        const uint32_t off = 0xfffffff0u | ((uint32_t)idx & 0u);
#else
        const uint32_t off = (uint32_t)idx * (uint32_t)sizeof(cplx<T>);
#endif
        if constexpr (sizeof(T) == 8) {
            using v4 = unsigned int __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4, v), rsrc, off, 0, 0);
        } else {
            using v2 = unsigned int __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2, v), rsrc, off, 0, 0);
        }
    }
};
__host__ __device__ inline int ceil_div_signed(int a, int b) {  // ceil(a / b), b > 0
    return a >= 0 ? (a + b - 1) / b : -((-a) / b);
}

template <bool WAVE>
__device__ inline void st_sync() {
#if defined(FV_ABL) && (FV_ABL & 4)  // diagnostic build: no barriers between the passes (wrong results; what they cost)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return;
#endif
    if constexpr (WAVE) {  // the row lives in one wavefront: LDS is in order, only the compiler must not reorder
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// v holds X[k] at v[bitrev(k)]: multiply X[k] by w^k, k = 1 .. R-1
template <typename T, int R>
__device__ inline void st_twiddle(cplx<T> *v, cplx<T> w) {
    constexpr int LR = ilog2_c(R);
    cplx<T> wk = w;
#pragma unroll
    for (int k = 1; k < R; ++k) {
        v[bitrev_small(k, LR)] = xmul(v[bitrev_small(k, LR)], wk);
        if (k + 1 < R) wk = xmul(wk, w);
    }
}

// the same with a common factor: X[k] *= start w^k, k = 0 .. R-1  (start = the residue twiddle w^{u p} of the thread's
// slots, which the radix-R butterfly commutes with: applying it here costs one multiply more than the plain pass-1
// twiddle, applying it to the R inputs cost R of them)
template <typename T, int R>
__device__ inline void st_twiddle_from(cplx<T> *v, cplx<T> start, cplx<T> w) {
    constexpr int LR = ilog2_c(R);
    cplx<T> wk = start;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        v[bitrev_small(k, LR)] = xmul(v[bitrev_small(k, LR)], wk);
        if (k + 1 < R) wk = xmul(wk, w);
    }
}
__device__ inline int mul24(int a, int b) { return (int)__mul24(a, b); }  // both factors below 2^23: full-rate multiply

// --- fused gather (small 2-D problems) ---------------------------------------------------------
// When the last FFT pass runs in column mode (8 adjacent lx per workgroup, all ly of one transform)
// and both dimensions are stored in natural order (P = 1), the workgroup can serve the targets
// directly from its LDS tile instead of writing C to memory for a separate gather kernel: every
// (target, frequency) item whose w footprint columns touch the workgroup's 8 columns receives that
// partial sum through two fp64 atomic adds (2-3 workgroups per item).  Everything that depends on
// (target, frequency) only -- footprint origin, the 2 w kernel weights, 1 / (psi_hat_x psi_hat_y),
// post-phase, flip -- is tabulated once per geometry (FgHdr records + per-(frequency, column group)
// item lists) and reused by every time step.
struct FgHdr {           // 48 bytes, followed by T kx[w], ky[w]
    int j0x, j0y;        // footprint origin (output indices)
    int64_t out_off;     // fg * out_fg_stride + k * out_k_stride
    double pr, pi;       // exp(i s . x_c) / (psi_hat_x psi_hat_y)
    double sgn;          // -1: conjugate (flipped baseline)
    double pad;
};
struct FusedArgs {
    const int *lstart;            // [nfg * ngx + 1] list offsets per (frequency, column group)
    const int *list;              // item ids
    const unsigned char *recs;    // records, rec bytes each
    int rec, ngx, tpol, w;
    void *out;                    // cplx<T> *, base of this (time, frequency-group) block
    void *out1;                   // the same for the second problem of a gang launch
    int64_t pol_off[4];
    int tflip;                    // flipped baselines land in the feed-transposed slot (InterpArgs::transpose_flipped)
};
struct FgGeom {
    int w, nox, noy, n2x, n2y;
    double hx, hy, btcx, btcy, xcx, xcy;
    int64_t out_fg_stride, out_k_stride;
    int rec, ngx;
};

// One thread per (target, frequency): record (BUILD) and the count / fill of the column-group lists.
template <typename T, int MODE>  // 0: records + counts, 1: fill lists
__global__ void k_fg_build(int64_t N, int nfg, const T *__restrict__ btx, const T *__restrict__ bty,
                           const int *__restrict__ bl_idx, const signed char *__restrict__ flip,
                           const double *__restrict__ scale, FgGeom gm, KerParams ker,
                           unsigned char *__restrict__ recs, int *__restrict__ counts,
                           const int *__restrict__ lstart, int *__restrict__ cursor, int *__restrict__ list) {
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= N * nfg) return;
    const int fg = (int)(item / N);
    FgHdr *h = reinterpret_cast<FgHdr *>(recs + item * gm.rec);
    if (MODE == 0) {
        const int64_t kl = item % N;
        const int64_t k = bl_idx ? bl_idx[kl] : kl;
        const double sg = (flip && flip[kl]) ? -1.0 : 1.0;
        const double sc = scale[fg];
        const T beta = (T)ker.beta, c4 = (T)ker.c;
        // same arithmetic as k_interp's preamble
        const double svx = sc * sg * (double)btx[k], svy = sc * sg * (double)bty[k];
        const double thx = gm.hx * (svx - sc * gm.btcx), thy = gm.hy * (svy - sc * gm.btcy);
        const double ex = thx * gm.n2x * (0.5 / M_PI) + 0.5 * gm.nox;
        const double ey = thy * gm.n2y * (0.5 / M_PI) + 0.5 * gm.noy;
        const int jx = max(0, min(gm.nox - gm.w, (int)ceil(ex - 0.5 * gm.w)));
        const int jy = max(0, min(gm.noy - gm.w, (int)ceil(ey - 0.5 * gm.w)));
        T *kx = reinterpret_cast<T *>(recs + item * gm.rec + sizeof(FgHdr));
        T *ky = kx + gm.w;
        for (int d = 0; d < gm.w; ++d) {
            kx[d] = es_eval<T>((T)((double)(jx + d) - ex), beta, c4);
            ky[d] = es_eval<T>((T)((double)(jy + d) - ey), beta, c4);
        }
        const double den = es_hat(ker, thx) * es_hat(ker, thy);
        const double ph = svx * gm.xcx + svy * gm.xcy;  // post-phase exp(i s . x_c)
        double pr = 1.0 / den, pi_ = 0.0;
        if (ph != 0.0) {
            double sn, cs;
            sincos(ph, &sn, &cs);
            pi_ = pr * sn;
            pr = pr * cs;
        }
        h->j0x = jx;
        h->j0y = jy;
        h->out_off = (int64_t)fg * gm.out_fg_stride + k * gm.out_k_stride;
        h->pr = pr;
        h->pi = pi_;
        h->sgn = sg;
        h->pad = 0.0;
        for (int g = jx >> 3; g <= (jx + gm.w - 1) >> 3; ++g) atomicAdd(&counts[fg * gm.ngx + g], 1);
    } else {
        const int jx = h->j0x;
        for (int g = jx >> 3; g <= (jx + gm.w - 1) >> 3; ++g) {
            const int b = fg * gm.ngx + g;
            list[lstart[b] + atomicAdd(&cursor[b], 1)] = (int)item;
        }
    }
}

// NLD = number of leading pass-1 operands that can be non-zero (n_in <= NLD Q/R1): the rest are
// compile-time zeros, which prunes the first butterfly stages.
// (Running the P residue jobs of a row group as one lock-stepped workgroup, with or without
// staging the merged row in LDS, was measured slower than separate workgroups on Q = 4096, P = 2.)
#ifndef FV_ST_DUAL
#define FV_ST_DUAL 1
#endif
// FOLD: rows longer than Q (n_in > Q; NLD = R1).  Slot q of the length-Q row then holds
//     w^{q p} sum_m x[q + m Q] c^m,      c = w^{Q p} = exp(2 pi i p / P)  (uniform),
// over the m with -n_in/2 <= q + m Q < n_in - n_in/2: ceil(n_in / Q) + 1 sweeps of coalesced loads, one
// uniform complex factor per sweep (none for p = 0), one per-slot twiddle at the end.
// INB (column mode): 0 = plain row-major input plane; 1 = column blocks of 2^in_blk elements (RowDifArgs::in_blk);
// 2 = column blocks exactly as wide as the workgroup's RPW columns.
// PAIR (folded rows, P >= 2): a job computes TWO residues (2 jp, 2 jp + 1) of its rows from ONE pass over their inputs.
// Why: the P residue jobs of a row all read the whole row, and although those re-reads hit in L2 it is the L2 / load
// path, not HBM and not the fp64 pipe, that bounds these kernels (in-kernel stamps: 50-63 % of a wave's life is the fold's
// load round trips at ~3 us each, the three butterflies together 5 %; VALU busy 45-50 %; C3's x-pass with P = 3 instead of
// 5 runs 40 % faster).  A thread has registers for ONE residue's 16 slots, so during the fold the two thread groups that
// will transform the two residues (g = 0, 1; same u) each take HALF of the slots and accumulate them for BOTH residues
// -- 8 + 8 accumulators, every loaded element used twice, half the loads per thread -- then hand the partner its
// residue's 8 slots through LDS (the exchange buffer is idle until pass 1) and continue, each with its own residue, exactly
// as an unpaired job.  Row mode: a workgroup's exchange rows are (row, residue) pairs; column mode: the workgroup keeps
// its columns (one 64-byte block) and doubles its threads.  Odd P: the last job's second group has no residue (it still
// folds its half of the slots for its partner, transforms garbage and stores nothing).
template <typename T, int LOGQ, bool COL, int NLD, bool FUSED = false, bool FOLD = false, int INB = 0, bool PAIR = false>
__global__ __launch_bounds__(st_threads(LOGQ, COL, PAIR), LOGQ == 12 && !COL ? FV_ST_MINW12 : 4) void k_rowfft_st(
    const cplx<T> *__restrict__ in0, cplx<T> *__restrict__ out0, const cplx<T> *__restrict__ tw, RowDifArgs a,
    FusedArgs fz) {
    static_assert(!FUSED || COL, "the fused gather rides on the column-mode last pass");
    static_assert(!FOLD || (!FUSED && NLD == (1 << (LOGQ == 9 ? 3 : 4))), "folding runs on full pass-1 operands");
    static_assert(!INB || COL, "blocked input planes are read by the column pass");
    static_assert(!PAIR || (FOLD && !FUSED && LOGQ >= 10), "paired residues: folded rows of Q >= 1024");
    // gang launch: blockIdx.y = 1 runs the same transform on a second pair of buffers
    const cplx<T> *__restrict__ in = blockIdx.y ? static_cast<const cplx<T> *>(a.in1) : in0;
    cplx<T> *__restrict__ out = blockIdx.y ? static_cast<cplx<T> *>(a.out1) : out0;
    using PL = StPlan<LOGQ, COL>;
    constexpr int R1 = PL::R1, R2 = PL::R2, R3 = PL::R3, TPR = PL::TPR, A = PL::A, B = PL::B;
    constexpr int THREADS = st_threads(LOGQ, COL, PAIR);
    constexpr int Q = 1 << LOGQ, S1 = R2 * R3, RPW = THREADS / TPR, ROW = PL::ROW;
    constexpr int NL = PAIR ? RPW / 2 : RPW;  // data lines (rows / columns) of a workgroup; RPW = its exchange rows
    static_assert(!PAIR || (RPW >= 2 && RPW % 2 == 0), "paired residues need an even number of exchange rows");
    static_assert(ROW >= R1 * A && A >= R2 * B && B >= R3, "LDS layout");
    constexpr int NI2 = R1 * R3 / TPR, NI3 = R1 * R2 / TPR;
    constexpr int L1 = ilog2_c(R1), L2 = ilog2_c(R2), L3 = ilog2_c(R3);
    constexpr bool WAVE = TPR == 64 && !COL;
    static_assert(S1 == TPR, "one pass-1 item per thread");
    // DUAL: separate buffers for the real and imaginary halves of an exchange, so that one barrier per
    // exchange suffices (column mode spreads a column's threads over all waves: its barriers are
    // workgroup-wide)
    constexpr bool DUAL = FV_ST_DUAL && COL && LOGQ == 9;
    // fp32 (CX): a complex value is 8 bytes -- the size of the fp64 real a slot holds -- so whole values cross the
    // exchange in ONE round (write, barrier, read) instead of two, on the bank-conflict-free strides that were searched
    // for 8-byte slots (StPlan); the halves of a value stay adjacent registers from the LDS read on, which the packed
    // arithmetic (PkF32) wants anyway.  OFF by default (FV_ST_CX32): it doubles the fp32 kernels' LDS to the fp64 kernels'
    // 33 KB per workgroup, which takes them from 7 to 4 waves per SIMD, and the lost occupancy costs more than the saved
    // round: C5's passes 1.47 -> 1.70 ms per launch (round 4, same box; round 3's compiler-packed attempt: 3.61 -> 4.10).
    constexpr bool CX = FV_ST_CX32 && sizeof(T) == 4 && !DUAL;
    using ST = std::conditional_t<CX, cplx<T>, T>;  // what an exchange slot holds
    __shared__ __attribute__((aligned(16))) ST smem[(DUAL ? 2 : 1) * RPW * ROW];
    // Pass-2 twiddles w_{Q/R1}^{j3 k2} = tw[j3 k2 P R1] (j3 < R3, k2 < R2) from a table in LDS, R3 R2 entries (2-4 KB),
    // filled from the global table at the start: the chain of R2 - 2 complex products per pass-2 item that formed the
    // powers in registers -- 56 of a thread's ~930 fp64 instructions per job at Q = 2048 -- becomes R2 - 1 reads.
    constexpr bool TW2 = FV_ST_TW2_LDS != 0;
    // (rows of R2 + 1 entries: the lanes of one LDS lane group read entry k of two or more different rows j3, and rows
    // R2 entries apart -- 256 B for fp64 -- sit on the same banks)
    constexpr int TW2S = R2 + 1;
    __shared__ __attribute__((aligned(16))) cplx<T> s_tw2[TW2 ? R3 * TW2S : 1];
    // Which pass-2 item (k1, j3) a thread takes is free (an item reads and rewrites its own R2 slots); the exchange slots
    // k1 A + j3 + n B of the items of 16 CONTIGUOUS lanes must fall into 16 different 8-byte slot classes mod 128 B --
    // that is how ds_read2_b64 / ds_write2_b64, which the compiler makes of the n-loops, are banked (4 groups of 16 lanes,
    // 32 banks) -- and the strides were searched for ds_read_b64's rule (2 x 32 lanes, 64 banks): with the natural
    // assignment k1 = v mod R1, j3 = v / R1 and A = 2 (mod 16) the 16 lanes hit 8 classes twice (PMC: 43 % of the row
    // passes' LDS cycles were bank-conflict cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  Row mode, R1 = 16: lane bits
    // (a = v mod 8, b = bit 3, c = v / 16) -> k1 = a + 8 (c mod 2), j3 = b + 2 (c / 2): classes 2 a + b (+ const).
    constexpr bool MAP2 = FV_ST_MAP2 && !COL && R1 == 16 && (A % 16) == 2 && sizeof(ST) == 8;
    auto item2 = [](int v) -> int {
        if constexpr (MAP2) {
            const int c = v >> 4;
            return ((v & 7) | ((c & 1) << 3)) + R1 * (((v >> 3) & 1) | ((c >> 1) << 1));
        } else {
            return v;
        }
    };

    const int tid = threadIdx.x;
    const int vb = blockIdx.x;
    if constexpr (TW2) {
        // (before the early exits: every thread that stays reads the table after at least one workgroup barrier)
        for (int e = tid; e < R3 * R2; e += THREADS) s_tw2[(e / R2) * TW2S + e % R2] = tw[mul24(mul24(e / R2, e % R2), a.P * R1)];
        if constexpr (WAVE) __syncthreads();  // rows of one wave synchronise by wave barriers only
    }
    // line = the data row / column of this thread inside the workgroup, g = its residue group (PAIR), r = its exchange row
    const int line = COL ? tid % NL : (PAIR ? (tid / TPR) >> 1 : tid / TPR);
    const int u = COL ? (tid / NL) % TPR : tid % TPR;
    const int g = !PAIR ? 0 : __builtin_amdgcn_readfirstlane(COL ? tid / (NL * TPR) : (tid / TPR) & 1);  // wave-uniform
    const int r = PAIR ? (COL ? g * NL + line : tid / TPR) : line;
#ifdef FV_FFT_STAMPS
    unsigned stamp_idx = 0xffffffffu;
    const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime();  // slot 8: the wave's first instruction; slot 9: its stores drained
    if ((tid & 63) == 0 && (FV_FFT_STAMPS != 2 || COL)) {  // FV_FFT_STAMPS = 2: column-mode kernels only
        stamp_idx = atomicAdd(&fv_stamp_count, 1u);
        // tag: kernel variant, HW_ID (wave slot 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13), XCC and block id: with them the
        // stamps also say how long a wave slot stays empty between two workgroups (scratch/stamps2.py)
        if (stamp_idx < (1u << 18))
            fv_stamps[(size_t)stamp_idx * 10 + 7] = ((unsigned long long)LOGQ << 8) | (COL ? 2 : 0) | (FOLD ? 1 : 0) | ((unsigned long long)(sizeof(T) == 8) << 2) | (PAIR ? 8 : 0) |
                                                    ((unsigned long long)(__builtin_amdgcn_s_getreg(63492) & 0xffffu) << 16) |
                                                    ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xfu) << 32) |
                                                    ((unsigned long long)(blockIdx.x & 0xffffffu) << 36);
    }
    FV_STAMP(0);
#endif
    const int PJ = PAIR ? (a.P + 1) >> 1 : a.P;  // jobs per line group
    int p;
    int64_t row0, row;
    if constexpr (COL) {
        // column mode: a workgroup owns NL adjacent columns and one residue (PAIR: two); the jobs of a column group get
        // block ids 8 apart (same XCD, back to back)
        const int tt = vb >> 3;
        p = tt % PJ;
        row0 = ((int64_t)(tt / PJ) * 8 + (vb & 7)) * NL;
        if (row0 >= a.nrows) return;  // workgroup-uniform
        row = row0 + line;
    } else {
        // row mode: XCD x (= block id mod 8) takes the x-th eighth of the rows, a CONTIGUOUS range -- neighbouring rows
        // write neighbouring 64-byte pieces of the blocked output, which then meet in one L2 (interleaving the row
        // groups over the XCDs cost the Q = 4096 x-pass 10 %) -- and within it jobs run row group by row group, the
        // jobs of a group back to back.  (Unpaired: the wave groups of a workgroup take different rows, never two
        // residues of one row: simultaneous requests for a line stall each other in the CU's L1 -- 1 506 vs 1 425 us.)
        const int i = vb >> 3, gq = i / PJ;
        p = i - gq * PJ;
        row0 = (int64_t)(vb & 7) * a.jobs_per_xcd + (int64_t)gq * NL;  // jobs_per_xcd: rows per XCD here
        if (gq * NL >= a.jobs_per_xcd || row0 >= a.nrows) return;  // workgroup-uniform
        row = row0 + line;
    }
    const int p_other = PAIR ? 2 * p + 1 - g : 0;  // the partner group's residue (may be P: no such residue)
    if constexpr (PAIR) p = 2 * p + g;             // this thread's residue from here on (wave-uniform); may be P
    const int64_t rplane = row / a.rpp, rk = row % a.rpp;
    const bool ok_line = row < a.nrows && rk < a.rpp_valid;  // the line exists: its inputs are read
    const bool ok = ok_line && p < a.P;                      // ... and so does this thread's residue: outputs are stored
    ST *rb = smem + r * ROW;
    ST *rbi = DUAL ? rb + RPW * ROW : rb;

    // fused gather: this 8-lane slot's first item (id, header, x weight) is requested now, so that
    // two of the three dependent round trips of the gather are long over when the tile is ready
    int fz_s1 = 0, fz_e = 0, fz_item = 0, fz_dx = -1;
    FgHdr fz_h{};
    T fz_wx = T(0);
    if constexpr (FUSED) {
        const int tr = (int)(row0 / a.rpp);
        const int b = (tr / fz.tpol) * fz.ngx + (int)((row0 % a.rpp) >> 3);
        fz_e = fz.lstart[b] + (tid >> 3);
        fz_s1 = fz.lstart[b + 1];
        if (fz_e < fz_s1) {
            fz_item = fz.list[fz_e];
            const unsigned char *rp = fz.recs + (int64_t)fz_item * fz.rec;
            fz_h = *reinterpret_cast<const FgHdr *>(rp);
            fz_dx = (int)(row0 % a.rpp) + (tid & 7) - fz_h.j0x;
            if ((unsigned)fz_dx < (unsigned)fz.w) fz_wx = reinterpret_cast<const T *>(rp + sizeof(FgHdr))[fz_dx];
        }
    }

    // ---- pass 1: load (+ twiddle / fold for the residue), radix R1, twiddle ---------------------
    // The transform is centred on input element hshift = n_in / 2: element ia enters at the cyclic
    // position s = ia - hshift (mod n2), i.e. the first hshift elements wrap to the END of the row.
    // exp(2 pi i (ia - hshift) l / n2) is then the plain DFT kernel of the shifted row and no output
    // needs a phase factor (an earlier version multiplied every output by w^{-hshift l}: a complex
    // multiply plus a step of the phase chain per output).  Slot q of the length-Q row holds the
    // elements with s = q (mod Q): one of them when n_in <= Q, else the extras are folded on top.
    cplx<T> va[R1];
    const int hshift = a.n_in / 2;
    // Input through a buffer descriptor.  Row mode: the row belongs to whole waves -- base = the row, extent = n_in,
    // the range check does the masking.  Column mode: base = the workgroup's first column, every lane adds its own
    // column's offset and masks by the index -1 (beyond any extent): one compare + select per element instead of
    // clamp + two compares + four selects, and no 64-bit addresses in vector registers.
    // Column mode, blocked input (INB): element (column x, row ia) of a plane sits at
    // (x >> b) (n_in << b) + (ia << b) + (x & (2^b - 1)), b = in_blk.  When the workgroup's RPW columns are exactly one
    // block (INB = 2: every fp64 pass, fp32 up to Q = 1024) that block is ONE contiguous run [ia][column] and a
    // descriptor over just that run does all the masking -- ia < 0 wraps to a huge offset, ia >= n_in lies past the
    // extent -- so an address is a single add; padding columns of the last block read allocated memory and are never
    // stored.  Otherwise one compare + select per element (the bound is 0 for a lane without a column), a shift or a
    // 24-bit multiply for the index -- no 32-bit integer multiplies (quarter rate) and no exec-mask regions.
    constexpr bool INBLK = INB != 0, ONEBLK = INB == 2;
    constexpr int BLK1 = ilog2_c(NL);  // the block width of INB = 2
    const int blk = ONEBLK ? BLK1 : a.in_blk;
    const int64_t plane0 = (row0 / a.rpp) * a.in_plane;
    const int64_t in_base = ONEBLK ? plane0 + ((((row0 % a.rpp) >> BLK1) * (int64_t)a.n_in) << BLK1)
                            : INBLK ? plane0
                            : COL   ? plane0 + (row0 % a.rpp) * a.in_row
                                    : (ok_line ? rplane * a.in_plane + rk * a.in_row : 0);
    // Row extents (row mode; RowDifArgs::row_ext): the row is non-zero, and was written, only in [xe0, xe1) -- the
    // descriptor covers exactly that piece (requests outside it return zero without touching memory) and the fold
    // below only makes the sweeps that can meet it.
    int xe0 = 0, xe1 = a.n_in;
    if constexpr (!COL) {
        if (a.row_ext) {
            const int eb = __builtin_amdgcn_readfirstlane((int)(rk >> 3)) * 2;
            xe0 = a.row_ext[eb];
            xe1 = a.row_ext[eb + 1];
        }
    }
    const RowBuf<T> rowin(in + in_base + (COL ? 0 : xe0), ONEBLK ? ((int64_t)a.n_in << BLK1) : COL ? -1 : (ok_line ? xe1 - xe0 : 0));
    const int lane_in = ONEBLK ? line
                        : INBLK ? (int)((rplane - row0 / a.rpp) * a.in_plane) + ((((int)rk >> blk) * a.n_in) << blk) +
                                      ((int)rk & ((1 << blk) - 1))
                        : COL   ? (int)(rplane * a.in_plane + rk * a.in_row - in_base)
                                : 0;
    const int in_elem = (int)a.in_elem;
    const unsigned nin_lane = ok_line ? (unsigned)a.n_in : 0u;
    auto in_index = [&](int ia) -> int {  // element index for the descriptor; out of its range where the row has no element
        if constexpr (ONEBLK)
            return (ia << BLK1) + lane_in;
        else if constexpr (INBLK)
            return (unsigned)ia < nin_lane ? lane_in + (ia << blk) : -1;
        else if constexpr (COL)
            return (unsigned)ia < nin_lane ? lane_in + mul24(ia, in_elem) : -1;
        else
            return ia - xe0;  // zero outside [xe0, xe1)
    };
    auto load_in = [&](int ia) -> cplx<T> { return rowin.load(in_index(ia)); };
    // Residue twiddle of slot q = u + k S1:  w^{q p} = w^{u p} (this thread's, one vector load) x w^{k S1 p} (uniform:
    // scalar loads).  Only the uniform part is applied to the inputs; w^{u p} is common to the thread's R1 slots, commutes
    // with their butterfly and rides on the pass-1 twiddle below (loaded there: no registers held across the loads and the butterfly).
    if constexpr (FOLD && PAIR) {
        static_assert(R1 == 16, "paired residues: 16 pass-1 slots per thread");
        const int lo_s = xe0 - hshift, nlo = xe1 - hshift;    // the row's elements sit at s in [lo_s, nlo)
        const int mmin = lo_s >> LOGQ;                        // floor(lo_s / Q)
        const int mmax = (nlo - 1) >> LOGQ;
        constexpr int CH = 4, HALF = R1 / 2;
        const int P = a.P;
        auto mod_p = [&](int v) {  // v mod P for v >= 0 (v < 4 P here)
            while (v >= P) v -= P;
            return v;
        };
        int mpm_min = (mmin * p) % P, mpo_min = (mmin * p_other) % P;  // (mmin p) mod P, mmin <= 0
        if (mpm_min < 0) mpm_min += P;
        if (mpo_min < 0) mpo_min += P;
        cplx<T> mine[HALF], theirs[HALF];  // this group's slots 8 g .. 8 g + 7, for its own residue and for the partner's
#pragma unroll
        for (int hh = 0; hh < HALF; hh += CH) {
            const int h = HALF * g + hh;  // first slot of the chunk (wave-uniform)
#pragma unroll
            for (int j = 0; j < CH; ++j) mine[hh + j] = theirs[hh + j] = {T(0), T(0)};
            const int m_lo = mmax - mmin < 2 ? mmin : max(mmin, -((((h + CH) * S1 - 1) - lo_s) >> LOGQ));  // ceil((lo_s - last slot) / Q)
            const int m_hi = mmax - mmin < 2 ? mmax : min(mmax, (nlo - 1 - h * S1) >> LOGQ);
            int mpm = mod_p(mpm_min + (m_lo - mmin) * mod_p(p)), mpo = mod_p(mpo_min + (m_lo - mmin) * mod_p(p_other));
            for (int m = m_lo; m <= m_hi; m += 2) {
                cplx<T> x[2][CH];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int off = (m + t) * Q + hshift;
#pragma unroll
                    for (int j = 0; j < CH; ++j) x[t][j] = load_in(m + t <= m_hi ? u + (h + j) * S1 + off : -1);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    // c^m of either residue (tw[0] = 1 exactly: no special case for a zero exponent)
                    const cplx<T> cm = tw[mpm * Q], co = tw[mpo * Q];
                    mpm = mod_p(mpm + mod_p(p));
                    mpo = mod_p(mpo + mod_p(p_other));
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        if constexpr (PkF32<T>::on) {
                            mine[hh + j] = pk_mac_s(mine[hh + j], x[t][j], cm);
                            theirs[hh + j] = pk_mac_s(theirs[hh + j], x[t][j], co);
                        } else {
                            mine[hh + j].re += x[t][j].re * cm.re - x[t][j].im * cm.im;
                            mine[hh + j].im += x[t][j].re * cm.im + x[t][j].im * cm.re;
                            theirs[hh + j].re += x[t][j].re * co.re - x[t][j].im * co.im;
                            theirs[hh + j].im += x[t][j].re * co.im + x[t][j].im * co.re;
                        }
                    }
                }
            }
            // the uniform part of the slot twiddle, w^{k S1 p}, k = h + j (k S1 p < n2 also for p = P)
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                mine[hh + j] = xmul_u(mine[hh + j], tw[(h + j) * S1 * p]);
                theirs[hh + j] = xmul_u(theirs[hh + j], tw[(h + j) * S1 * p_other]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // hand the partner group (same line, same u, other residue) its 8 slots through the idle exchange buffer:
        // [exchange row][slot][u] complex, lanes contiguous (column mode: [group][slot][u][column])
        {
            cplx<T> *xb = reinterpret_cast<cplx<T> *>(smem);
            static_assert(HALF * TPR * 2 <= ROW, "partner hand-over fits the exchange rows");
            const int mybase = COL ? (g * HALF * TPR + u) * NL + line : (r * HALF) * TPR + u;
            const int pabase = COL ? ((1 - g) * HALF * TPR + u) * NL + line : ((r ^ 1) * HALF) * TPR + u;
            constexpr int KS = COL ? TPR * NL : TPR;  // stride of a slot
#pragma unroll
            for (int kk = 0; kk < HALF; ++kk) xb[mybase + kk * KS] = theirs[kk];
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < HALF; ++kk) theirs[kk] = xb[pabase + kk * KS];  // now: the partner's slots of MY residue
            __syncthreads();  // read before the exchanges overwrite it
        }
        // odd P: the last job's second group has folded its half for the partner and has no residue of its own --
        // its waves end here (whole waves: a barrier counts the waves still running)
        if (p >= a.P) return;
        if (g == 0) {  // wave-uniform: slots 0..7 are mine, 8..15 came from the partner
#pragma unroll
            for (int kk = 0; kk < HALF; ++kk) {
                va[kk] = mine[kk];
                va[HALF + kk] = theirs[kk];
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < HALF; ++kk) {
                va[HALF + kk] = mine[kk];
                va[kk] = theirs[kk];
            }
        }
    } else if constexpr (FOLD) {
        const int lo_s = (COL ? 0 : xe0) - hshift, nlo = (COL ? a.n_in : xe1) - hshift;  // the row's elements sit at s in [lo_s, nlo)
        const int mmin = lo_s >> LOGQ;                        // floor(lo_s / Q)
        const int mmax = (nlo - 1) >> LOGQ;
        constexpr int CH = 4;  // slots per chunk: 4 accumulators + 2 sweeps x 4 operands in flight beside va
        int mp_min = (mmin * p) % a.P;  // (mmin p) mod P, mmin <= 0
        if (mp_min < 0) mp_min += a.P;
#pragma unroll
        for (int h = 0; h < R1; h += CH) {
            cplx<T> acc[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) acc[j] = {T(0), T(0)};
            // sweeps that can hold an element for SOME slot of this chunk (slots h S1 .. (h + CH) S1 - 1 over the row's
            // threads): the outermost sweeps only reach the first / last chunk -- uniform bounds, so the others
            // neither load nor accumulate them (n_in = 4688, Q = 2048: 3, 2, 2, 3 sweeps instead of 4 each)
            // (rows of two sweeps have nothing to skip, and their loop runs 4 % faster on the plain bounds)
            const int m_lo = mmax - mmin < 2 ? mmin : max(mmin, -((((h + CH) * S1 - 1) - lo_s) >> LOGQ));  // ceil((lo_s - last slot) / Q)
            const int m_hi = mmax - mmin < 2 ? mmax : min(mmax, (nlo - 1 - h * S1) >> LOGQ);
            // two sweeps per round trip: 8 loads in flight (a sweep beyond m_hi asks for index -1: zero, no access).
            // (Two more sweeps per trip by LDS-DMA into the idle exchange buffer -- no registers -- were measured to
            // change nothing: the job is bound by the throughput of the L2 / load path that serves the P re-reads of
            // every row, not by its latency; see PAIR below.)
            int mpw = mp_min + (m_lo - mmin) * p;  // (m_lo p) mod P: m_lo - mmin is 0 or 1
            while (mpw >= a.P) mpw -= a.P;
            auto accumulate = [&](const cplx<T> *xs) {
                // c^m = w^{(m Q p) mod n2}: m Q p is a multiple of Q, so the index is ((m p) mod P) Q -- walked
                // incrementally (a modulo by a run-time P is a dozen scalar instructions, and this is per sweep)
                const int mp = mpw;
                mpw += p;
                if (mpw >= a.P) mpw -= a.P;
                if (mp) {  // uniform
                    const cplx<T> cm = tw[mp * Q];
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        if constexpr (PkF32<T>::on) {
                            acc[j] = pk_mac_s(acc[j], xs[j], cm);
                        } else {
                            acc[j].re += xs[j].re * cm.re - xs[j].im * cm.im;
                            acc[j].im += xs[j].re * cm.im + xs[j].im * cm.re;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        if constexpr (PkF32<T>::on) {
                            acc[j] = pk_add(acc[j], xs[j]);
                        } else {
                            acc[j].re += xs[j].re;
                            acc[j].im += xs[j].im;
                        }
                    }
                }
            };
            // sweeps per round trip: two (8 loads in flight), or -- row mode, fp64 -- three where a chunk has exactly three
            // sweeps to make (rows of 2 Q < n_in < 3 Q: the first and last chunk of C3's 4 688-long rows at Q = 2048),
            // which then take one round trip instead of two (12 loads in flight; the chunks' results va[] are still few
            // when the first chunk runs, and the last chunk is the one register the allocation is sized by anyway)
            constexpr int NSW = FV_ST_FOLD_SWEEPS3 && !COL && sizeof(T) == 8 ? 3 : 2;
            if (NSW == 3 && m_hi - m_lo == 2) {
                cplx<T> x[3][CH];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int off = (m_lo + t) * Q + hshift;
#pragma unroll
                    for (int j = 0; j < CH; ++j) x[t][j] = load_in(u + (h + j) * S1 + off);
                }
                accumulate(x[0]);
                accumulate(x[1]);
                accumulate(x[2]);
            } else
            for (int m = m_lo; m <= m_hi; m += 2) {
                cplx<T> x[2][CH];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int off = (m + t) * Q + hshift;
#pragma unroll
                    for (int j = 0; j < CH; ++j) x[t][j] = load_in(m + t <= m_hi ? u + (h + j) * S1 + off : -1);
                }
                accumulate(x[0]);
                accumulate(x[1]);
            }
            if (p) {
                // the uniform part of the slot twiddle, w^{k S1 p} (k = 0: one)
#pragma unroll
                for (int j = 0; j < CH; ++j)
                    if (h + j) acc[j] = xmul_u(acc[j], tw[(h + j) * S1 * p]);  // k S1 p < n2
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) va[h + j] = acc[j];
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const int nlo = a.n_in - hshift;  // elements with s >= 0
        constexpr int NH = NLD / 2;  // NLD < R1: only the first and the last NH registers can be non-zero
        // (the Q = 4096 row kernel is allowed 168 registers: one round trip instead of two)
        if constexpr (LOGQ == 12 && !COL && NLD == R1) {
            // all NLD loads of the thread are issued back to back, straight into their pass-1 registers (branch-free:
            // the descriptor / an index of -1 masks); the residue twiddles follow in chunks of 4 and are multiplied in
            auto slot = [&](int i, int &jr, int &q, bool &hi) {
                jr = NLD == R1 ? i : (i < NH ? i : R1 - NLD + i);
                q = u + jr * S1;
                // slot q: the element with s = q if there is one, else the wrapped one with s = q - Q
                hi = NLD == R1 ? q >= nlo : i >= NH;
            };
    #pragma unroll
            for (int i = 0; i < NLD; ++i) {
                int jr, q;
                bool hi;
                slot(i, jr, q, hi);
                va[jr] = load_in((hi ? q - Q : q) + hshift);  // zero where the row has no element
            }
            if (p) {  // workgroup-uniform
                // uniform part of the residue twiddle: slot k holds s = u + k S1 (twiddle w^{k S1 p}) or the wrapped
                // element s = u + k S1 - Q (w^{k S1 p} conj(c), c = w^{Q p}); which one is a per-lane select between two
                // uniform values when the row is long (NLD == R1), a compile-time fact otherwise
                const cplx<T> cq = tw[Q * p];  // Q p < n2 for p < P
    #pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    int jr, q;
                    bool hi;
                    slot(i, jr, q, hi);
                    const cplx<T> f = tw[jr * S1 * p];
                    const cplx<T> fc = cmul(f, cplx<T>{cq.re, -cq.im});
                    if (NLD == R1)
                        va[jr] = xmul(va[jr], cplx<T>{hi ? fc.re : f.re, hi ? fc.im : f.im});
                    else
                        va[jr] = xmul(va[jr], i >= NH ? fc : f);
                }
            }
        } else {
            // branch-free loads (clamped index, masked afterwards) so that a chunk's requests issue
            // back to back; chunks of 8 bound the registers held by data + residue twiddles in flight
            constexpr int CH = NLD < 8 ? NLD : 8;
    #pragma unroll
            for (int h = 0; h < NLD; h += CH) {
                cplx<T> x[CH];
                bool his[CH];
    #pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int jr = NLD == R1 ? h + j : (h + j < NH ? h + j : R1 - NLD + h + j);
                    const int q = u + jr * S1;
                    // slot q: the element with s = q if there is one, else the wrapped one with s = q - Q
                    const bool hi = NLD == R1 ? q >= nlo : h + j >= NH;
                    his[j] = hi;
                    x[j] = load_in((hi ? q - Q : q) + hshift);
                }
                if (p) {  // workgroup-uniform: the uniform part of the residue twiddle, w^{k S1 p} (wrapped: x conj(w^{Q p}))
                    const cplx<T> cq = tw[Q * p];  // Q p < n2 for p < P
    #pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int jr = NLD == R1 ? h + j : (h + j < NH ? h + j : R1 - NLD + h + j);
                        const cplx<T> f = tw[jr * S1 * p];
                        const cplx<T> fc = cmul(f, cplx<T>{cq.re, -cq.im});
                        if (NLD == R1)
                            x[j] = xmul(x[j], cplx<T>{his[j] ? fc.re : f.re, his[j] ? fc.im : f.im});
                        else
                            x[j] = xmul(x[j], h + j >= NH ? fc : f);
                    }
                }
    #pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int jr = NLD == R1 ? h + j : (h + j < NH ? h + j : R1 - NLD + h + j);
                    va[jr] = x[j];  // already zero where the row has no element
                }
                if (h + CH < NLD) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (NLD < R1) {
#pragma unroll
            for (int n1 = NH; n1 < R1 - NH; ++n1) va[n1] = {T(0), T(0)};
        }
    }
    FV_STAMP(1);  // inputs loaded (and folded)
    dif_regs<T, R1>(va);
    st_twiddle_from<T, R1>(va, tw[mul24(u, p)], tw[mul24(u, a.P)]);  // w^{u p} w_Q^{u k1}, w_Q = w_{n2}^P; u p < n2 (p = 0: tw[0] = 1)
    const int s1 = (u / R3) * B + (u % R3);

    int base2[NI2], base3[NI3];
#pragma unroll
    for (int i = 0; i < NI2; ++i) {
        const int v = item2(u + i * TPR);
        base2[i] = (v % R1) * A + (v / R1);
    }
#pragma unroll
    for (int i = 0; i < NI3; ++i) {
        const int v = u + i * TPR;
        base3[i] = (v % R1) * A + (v / R1) * B;
    }

    // ---- exchange 1 -> pass 2 -------------------------------------------------------------------
    FV_STAMP(2);  // pass 1 done
    cplx<T> vb2[NI2][R2];
    if constexpr (CX) {
#pragma unroll
        for (int k = 0; k < R1; ++k) rb[s1 + k * A] = va[bitrev_small(k, L1)];
        st_sync<WAVE>();
#pragma unroll
        for (int i = 0; i < NI2; ++i)
#pragma unroll
            for (int n = 0; n < R2; ++n) vb2[i][n] = rb[base2[i] + n * B];
    } else {
#pragma unroll
    for (int k = 0; k < R1; ++k) rb[s1 + k * A] = va[bitrev_small(k, L1)].re;
    if constexpr (!DUAL) {
        st_sync<WAVE>();
#pragma unroll
        for (int i = 0; i < NI2; ++i)
#pragma unroll
            for (int n = 0; n < R2; ++n) vb2[i][n].re = rb[base2[i] + n * B];
        st_sync<WAVE>();
    }
#pragma unroll
    for (int k = 0; k < R1; ++k) rbi[s1 + k * A] = va[bitrev_small(k, L1)].im;
    st_sync<WAVE>();
    if constexpr (DUAL) {
#pragma unroll
        for (int i = 0; i < NI2; ++i)
#pragma unroll
            for (int n = 0; n < R2; ++n) vb2[i][n].re = rb[base2[i] + n * B];
    }
#pragma unroll
    for (int i = 0; i < NI2; ++i)
#pragma unroll
        for (int n = 0; n < R2; ++n) vb2[i][n].im = rbi[base2[i] + n * B];
    }

    FV_STAMP(3);  // exchange 1 done
#pragma unroll
    for (int i = 0; i < NI2; ++i) {
        const int j3 = item2(u + i * TPR) / R1;
        dif_regs<T, R2>(vb2[i]);
        if constexpr (TW2) {
#pragma unroll
            for (int k = 1; k < R2; ++k) vb2[i][bitrev_small(k, L2)] = xmul(vb2[i][bitrev_small(k, L2)], s_tw2[j3 * TW2S + k]);
        } else {
            st_twiddle<T, R2>(vb2[i], tw[mul24(j3, a.P * R1)]);  // w_{Q/R1}^{j3 k2}
        }
    }

    // ---- exchange 2 -> pass 3 (pass-2 items write back to the slots they read: no sync before) --
    FV_STAMP(4);  // pass 2 done
    cplx<T> vc[NI3][R3];
    if constexpr (CX) {
#pragma unroll
        for (int i = 0; i < NI2; ++i)
#pragma unroll
            for (int k = 0; k < R2; ++k) rb[base2[i] + k * B] = vb2[i][bitrev_small(k, L2)];
        st_sync<WAVE>();
#pragma unroll
        for (int i = 0; i < NI3; ++i)
#pragma unroll
            for (int n = 0; n < R3; ++n) vc[i][n] = rb[base3[i] + n];
    } else {
#pragma unroll
    for (int i = 0; i < NI2; ++i)
#pragma unroll
        for (int k = 0; k < R2; ++k) rb[base2[i] + k * B] = vb2[i][bitrev_small(k, L2)].re;
    if constexpr (!DUAL) {
        st_sync<WAVE>();
#pragma unroll
        for (int i = 0; i < NI3; ++i)
#pragma unroll
            for (int n = 0; n < R3; ++n) vc[i][n].re = rb[base3[i] + n];
        st_sync<WAVE>();
    }
#pragma unroll
    for (int i = 0; i < NI2; ++i)
#pragma unroll
        for (int k = 0; k < R2; ++k) rbi[base2[i] + k * B] = vb2[i][bitrev_small(k, L2)].im;
    st_sync<WAVE>();
    if constexpr (DUAL) {
#pragma unroll
        for (int i = 0; i < NI3; ++i)
#pragma unroll
            for (int n = 0; n < R3; ++n) vc[i][n].re = rb[base3[i] + n];
    }
#pragma unroll
    for (int i = 0; i < NI3; ++i)
#pragma unroll
        for (int n = 0; n < R3; ++n) vc[i][n].im = rbi[base3[i] + n];
    }

    // ---- pass 3 and this residue's outputs: k' = v + k3 Q/R3, l = P k' + p (mod n2, signed) ------
    FV_STAMP(5);  // exchange 2 done
    if (!FUSED && !ok) return;
    const int half_n = a.n_out / 2;
    // residue-major storage: l = P ks + p sits at ((p + half_n) mod P) cnt + (p + half_n) / P + ks
    // (natural order when cnt == 0: position l + half_n = base + P ks)
    cplx<T> *rout = out + (rplane * a.rpp_valid + rk) * a.out_pitch;
    rout += a.cnt ? ((p + half_n) % a.P) * a.cnt + (p + half_n) / a.P : p + half_n;
    int ostep = a.cnt ? 1 : a.P;
    if constexpr (FUSED) {  // P = 1, natural order: the column's outputs go to the LDS tile [RPW][n_out]
        __syncthreads();    // every pass-3 operand has left the exchange buffers
        rout = reinterpret_cast<cplx<T> *>(smem) + (int64_t)r * a.n_out + half_n;
        ostep = 1;
    }
    // Output through a buffer descriptor.  Row mode: exactly this residue's run of outputs, ks in [ks_lo, ks_hi):
    // stores outside it are dropped by the range check -- no compare, no exec masking, no branch per output.
    // Column mode: base = the first column's run, lanes add their own row's offset and mask by the index -1.
    const int ks_lo = ceil_div_signed(-half_n - p, a.P), ks_hi = ceil_div_signed(a.n_out - half_n - p, a.P);
    const int64_t res_off = (a.cnt ? ((p + half_n) % a.P) * a.cnt + (p + half_n) / a.P : p + half_n) + (int64_t)ks_lo * ostep;
#if defined(FV_ABL) && (FV_ABL & 8)  // diagnostic build: the x-pass stores contiguous runs (the y-pass then reads garbage)
    const bool out_blocked = false;
#else
    const bool out_blocked = !COL && !FUSED && a.out_blk;
#endif
    const int64_t out_base = out_blocked ? rplane * a.rpp_valid * a.out_pitch + (rk << a.out_blk)
                             : COL       ? ((row0 / a.rpp) * a.rpp_valid + row0 % a.rpp) * a.out_pitch + res_off
                                         : (rplane * a.rpp_valid + rk) * a.out_pitch + res_off;
    const RowBuf<T> rowout(out + out_base,
                           FUSED ? 0 : (COL || out_blocked) ? -1 : (!ok || ks_hi <= ks_lo ? 0 : (int64_t)(ks_hi - ks_lo - 1) * ostep + 1));
    const int blk_rows = (int)a.rpp_valid << a.out_blk, blk_mask = (1 << a.out_blk) - 1, res0 = (int)res_off;
    const int blk_step = (((Q / R3) * ostep) >> a.out_blk) * blk_rows, blk_wrap = ((Q * ostep) >> a.out_blk) * blk_rows;
    const unsigned blk_len = ok && ks_hi > ks_lo ? (unsigned)(ks_hi - ks_lo) : 0u;
    const int lane_out = COL ? (int)((rplane * a.rpp_valid + rk) * a.out_pitch + res_off - out_base) : 0;
    unsigned long long om_lo = ~0ull, om_hi = ~0ull;  // output mask of this (frequency, column block, residue): uniform
    if constexpr (COL && !FUSED && LOGQ <= 11) {
        if (a.omask) {
            constexpr int NW = Q > 1024 ? Q / 1024 : 1;
            const int64_t fgi = (row0 / a.rpp) / a.omask_tpol, bi = (row0 % a.rpp) / NL;
            const unsigned long long *mp = a.omask + ((fgi * a.omask_nblk + bi) * a.P + p) * NW;
            om_lo = mp[0];
            om_hi = NW > 1 ? mp[NW - 1] : 0ull;
        }
    }
    const int col_step = (Q / R3) * ostep, col_wrap = Q * ostep;  // uniform
    int col_pos0[NI3];
#pragma unroll
    for (int i = 0; i < NI3; ++i) col_pos0[i] = lane_out + mul24(u + i * TPR - ks_lo, ostep);
    // Column plan: the compact column of each of this thread's outputs, requested now -- one 4-byte load per output from
    // a table row of a few KiB that every row of the transform reads -- so that the answers are back when pass 3 ends.
    // A descriptor over the table row masks the outputs beyond this residue's run (index -1: zero = "not stored").
    bool use_ctab = false;
    int ct[NI3][R3];
    if constexpr (!COL && !FUSED) {
        use_ctab = out_blocked && a.ctab != nullptr && ostep == 1;
        if (use_ctab) {
            // a descriptor over exactly this residue's run of the table row: entry `rel` of the run, and zero -- "not
            // stored" -- for the outputs beyond it, without a compare (negative rel wraps past the extent)
            const int *trow = wave_uniform_ptr(a.ctab + (int64_t)__builtin_amdgcn_readfirstlane((int)(rplane / a.ctab_tpol)) * a.ctab_stride +
                                               __builtin_amdgcn_readfirstlane(res0));
            const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(trow), 0, (uint32_t)__builtin_amdgcn_readfirstlane((int)blk_len) * 4u, 0x00020000);
#pragma unroll
            for (int i = 0; i < NI3; ++i) {
                const int rel0 = (u + i * TPR - ks_lo) * 4;
#pragma unroll
                for (int k = 0; k < R3; ++k)
                    ct[i][k] = __builtin_amdgcn_raw_buffer_load_b32(trs, (uint32_t)(rel0 + (k * (Q / R3) - (k >= R3 / 2 ? Q : 0)) * 4), 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NI3; ++i) {
        int v = u + i * TPR;
        // opaque to the optimiser: the output indices v + k Q/R3 are the same numbers as pass 1's slot indices,
        // and kept alive from there they cost 16 registers for three passes (recomputing is one add each)
        asm volatile("" : "+v"(v));
        dif_regs<T, R3>(vc[i]);
        // Only about half of a residue's Q outputs are wanted (|l| <= n_out / 2): for most (wave, k) the whole store would
        // be dropped by the range check -- after being issued.  The wave's lanes hold consecutive v, so whether ANY lane's
        // output k is wanted is a scalar test: the dead stores (a third of the FFT passes' time went into stores) are
        // branched over.
        const int v_first = __builtin_amdgcn_readfirstlane(v), v_span = COL ? 64 / NL - 1 : 63;
        const int want = ks_hi - ks_lo;  // uniform
#pragma unroll
        for (int k = 0; k < R3; ++k) {
            const int kk = v + k * (Q / R3);
            const int ks = kk < Q / 2 ? kk : kk - Q;
            const int l = a.P * ks + p;
            if constexpr (!FUSED && !(COL && LOGQ == 12)) {  // (the 1024-thread column kernel has no register to spare)
                const int rel_first = (v_first - ks_lo) + k * (Q / R3) - (k >= R3 / 2 ? Q : 0);
                if (rel_first + v_span < 0 || rel_first >= want) continue;  // wave-uniform
                if constexpr (COL && LOGQ <= 11) {  // column plan: chunks of 16 outputs no target reads
                    const int c = (v_first + k * (Q / R3)) >> 4;
                    if (!(((c < 64 ? om_lo : om_hi) >> (c & 63)) & 1ull)) continue;  // wave-uniform
                }
            }
            if constexpr (FUSED) {
                if (l >= -half_n && l < a.n_out - half_n) rout[ks * ostep] = vc[i][bitrev_small(k, L3)];
            } else if constexpr (COL) {
                // ks wraps exactly at k = R3 / 2 (v < Q / R3): position and validity are affine in k
                const int rel = (v - ks_lo) + k * (Q / R3) - (k >= R3 / 2 ? Q : 0);
                rowout.store((unsigned)rel < blk_len ? col_pos0[i] + k * col_step - (k >= R3 / 2 ? col_wrap : 0) : -1,
                             vc[i][bitrev_small(k, L3)]);
            } else {
                if (out_blocked) {
                    // a thread's outputs are whole blocks apart (Q / R3 is a multiple of the block width), and
                    // ks wraps exactly at k = R3 / 2 (v < Q / R3): index and validity are affine in k with
                    // uniform constants -- two adds, a compare and a select per store
                    const int rel = (v - ks_lo) + k * (Q / R3) - (k >= R3 / 2 ? Q : 0);
                    if (use_ctab) {  // uniform: compact columns, only those a target reads
                        rowout.store(ct[i][k] - 1, vc[i][bitrev_small(k, L3)]);  // 0: index -1, dropped by the range check
                        continue;
                    }
                    const int pos0 = res0 + mul24(v - ks_lo, ostep);
                    const int idx = mul24(pos0 >> a.out_blk, blk_rows) + (pos0 & blk_mask) + k * blk_step - (k >= R3 / 2 ? blk_wrap : 0);
                    rowout.store((unsigned)rel < blk_len ? idx : -1, vc[i][bitrev_small(k, L3)]);
                } else {
                    // outside the residue's run the index is negative or past the extent: dropped by the descriptor
                    rowout.store(mul24(v - ks_lo, ostep) + k * col_step - (k >= R3 / 2 ? col_wrap : 0), vc[i][bitrev_small(k, L3)]);
                }
            }
        }
    }
    FV_STAMP(6);  // pass 3 done, stores issued
    if constexpr (FUSED) {
        // ---- gather from the tile: 8 lanes (one per column) per item, 64 items in flight -----------
        __syncthreads();
        const cplx<T> *tile = reinterpret_cast<const cplx<T> *>(smem);
        const int64_t rk0 = row0 % a.rpp;                    // first column of the workgroup
        const int pol = (int)(row0 / a.rpp) % fz.tpol;       // transform = (frequency, polarisation)
        const int c = tid & 7;
        cplx<T> *obase0 = reinterpret_cast<cplx<T> *>(blockIdx.y ? fz.out1 : fz.out);
        const int64_t po_plain = fz.pol_off[pol];
        const int64_t po_flip = fz.tflip && fz.tpol == 4 ? fz.pol_off[(pol & 1) * 2 + (pol >> 1)] : po_plain;
        for (int e = fz_e; e < fz_s1; e += st_threads(LOGQ, COL) / 8) {
            if (e != fz_e) {  // beyond the prefetched first round
                fz_item = fz.list[e];
                const unsigned char *rp = fz.recs + (int64_t)fz_item * fz.rec;
                fz_h = *reinterpret_cast<const FgHdr *>(rp);
                fz_dx = (int)rk0 + c - fz_h.j0x;
                fz_wx = (unsigned)fz_dx < (unsigned)fz.w ? reinterpret_cast<const T *>(rp + sizeof(FgHdr))[fz_dx] : T(0);
            }
            T sr = T(0), si = T(0);
            if ((unsigned)fz_dx < (unsigned)fz.w) {
                const T *ky = reinterpret_cast<const T *>(fz.recs + (int64_t)fz_item * fz.rec + sizeof(FgHdr)) + fz.w;
                const cplx<T> *col = tile + (int64_t)c * a.n_out + fz_h.j0y;
                for (int d = 0; d < fz.w; ++d) {
                    const T wy = ky[d];
                    sr += col[d].re * wy;
                    si += col[d].im * wy;
                }
                sr *= fz_wx;
                si *= fz_wx;
            }
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) {
                sr += __shfl_xor(sr, off, 64);
                si += __shfl_xor(si, off, 64);
            }
            if (c == 0) {
                const double vr = (double)sr * fz_h.pr - (double)si * fz_h.pi;
                const double vi = ((double)sr * fz_h.pi + (double)si * fz_h.pr) * fz_h.sgn;
                cplx<T> *o = obase0 + (fz_h.sgn < 0 ? po_flip : po_plain) + fz_h.out_off;
                atomicAdd(&o->re, (T)vr);
                atomicAdd(&o->im, (T)vi);
            }
        }
    }
}

// [batch][R][C] -> [batch][C][R], 32x32 tiles through LDS (both sides coalesced).
template <typename T>
__global__ void k_transpose(const cplx<T> *__restrict__ in, cplx<T> *__restrict__ out, int R, int C) {
    __shared__ cplx<T> tile[32][33];
    const int64_t plane = (int64_t)blockIdx.z * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per sweep
    for (int rr = ty; rr < 32; rr += 8) {
        const int rI = r0 + rr, cI = c0 + tx;
        if (rI < R && cI < C) tile[rr][tx] = in[plane + (int64_t)rI * C + cI];
    }
    __syncthreads();
    for (int cc = ty; cc < 32; cc += 8) {
        const int cO = c0 + cc, rO = r0 + tx;
        if (cO < C && rO < R) out[plane + (int64_t)cO * R + rO] = tile[tx][cc];
    }
}

// --- gather (interp), 2-D and 3-D ---------------------------------------------------------------
// The transformed grid arrives with its dimensions reversed by the FFT passes: 2-D [trans][lx][ly],
// 3-D [trans][lx][ly][lz].  Index 0 of the per-dimension arrays below is the contiguous ("fast")
// dimension (y in 2-D, z in 3-D), index 1 the next one, index 2 (3-D only) the slowest.
struct InterpArgs {
    int w, tpol, nfg;             // tpol transforms per frequency group, nfg groups
    int64_t items_per_xcd;        // (frequency, target) items per XCD: see k_interp
    int n2[3], no[3];
    int P[3], cnt[3];             // residue-major column storage (DimGeom::out_pos), row length P cnt
    double h[3];                  // theta = h * s'
    double btc[3], xc[3];
    int64_t out_fg_stride;        // output element strides
    int64_t out_k_stride;
    int64_t out_pol_off[16];      // offset of polarisation product r (r < tpol <= 16); beyond: r * out_pol_off[1]
    int accumulate;               // out += instead of out =
    int herm;                     // 0 | 1 Hermitian strengths | 2 all-real strengths (k_interp<.., HERM>)
    // eigenbeam contraction (cpu_simulate.py:461-468): when basis != 0 every value is added as
    //   conj(C[a1,kk,f]) C[a2,ll,f] V_r            at polarisation slot r, and (kk != ll) as
    //   conj(C[a1,ll,f]) C[a2,kk,f] V_r            at the feed-transposed slot.
    int basis, kk, ll, nbasis, ncoef_freq, f_first;
    // reference_compat = 0 (the reference's forms are the default):
    //   transpose_flipped: a flipped baseline of a two-beam pair is V_ij(-b)^H -- conjugated (as the reference does,
    //   cpu_simulate.py:298) AND written to the feed-transposed slot;
    //   basis_part: 1 = add only the (kk, ll) term, 2 = add only the transposed (ll, kk) term (from THIS launch's
    //   values: with negate_all they are conj(V_kl(-b)), the exact V_lk(b)^T for complex basis beams), 0 = both
    //   from V_kl(b) (cpu_simulate.py:464-468);  negate_all: every target is taken at -s and conjugated.
    int transpose_flipped, basis_part, negate_all;
    // Column plan (2-D; Nufft3::arm_columns): the slow dimension of the grid holds only the ncc columns some target
    // reads; ctab[fg ctab_stride + position] = 1 + compact column (0: left out -- never met by a footprint the plan was
    // built from; counted in *err if it happens).
    const int *ctab;
    int ctab_stride, ncc;
    int *err;
    // Height term of a nearly flat array (Sim::run, "w-term expansion"): this launch gathered the 2-D transform F_k of the
    // strengths c_j ((z_j - wt_zc) / wt_zh)^k, and every member adds  exp(i wt_zc zq) (i wt_zh zq)^k / k!  F_k  with its own
    // sign-adjusted height coordinate zq = nu (+-b_z) (wt_bz: the baselines' third component, by global baseline id) --
    // the k-th term of exp(i z_j zq) expanded about the middle of the sources' height range.  wt_k < 0: off.
    int wt_k;
    double wt_zc, wt_zh;
    const void *wt_bz;
    // Direct third dimension (Nufft3::zdirect; the 2-D instantiations): the grid holds, per transform, zd_n (x, y)
    // transforms -- one per z-plane of the spread grid, plane k at z_k = xc_z + (k - zd_n / 2) h_z -- and the target sums
    // them with their exact phases,  sum_k exp(i (k - zd_n / 2) theta_z) G_k(s_x, s_y),  theta_z = h_z (s_z - s_c,z),
    // divided by the kernel's transform at theta_z like every dimension: no transform pass and no interpolation along z
    // (bt2 = the targets' third coordinate).  0: off.
    int zd_n;
    double zd_h, zd_btc, zd_xc;
};

// Bessel function of the first kind and integer order k >= 0 by its power series, (x / 2)^k / k! sum_m (-x^2 / 4)^m /
// (m! (k + 1)_m): the arguments here are the height phases of a nearly flat array (|x| below ~4 while the expansion is
// taken at all), where the series loses at most a digit to cancellation and needs a dozen terms.
__device__ inline double bessel_j_series(int k, double x) {
    double lead = 1.0;
    for (int i = 1; i <= k; ++i) lead *= 0.5 * x / (double)i;
    const double q = -0.25 * x * x;
    double term = 1.0, sum = 1.0;
    for (int m = 1; m < 60; ++m) {
        term *= q / ((double)m * (double)(m + k));
        sum += term;
        if (fabs(term) < 1e-17 * fabs(sum)) break;
    }
    return lead * sum;
}

// exp(i zc zq) c_k(zh zq),  c_0 = J_0, c_k = 2 i^k J_k: the Chebyshev (Jacobi - Anger) coefficients of
// exp(i a t) = J_0(a) + 2 sum_k i^k J_k(a) T_k(t) on t in [-1, 1]
__device__ inline cplx<double> wterm_factor(int k, double zc, double zh, double zq) {
    double sn, cs;
    sincos(zc * zq, &sn, &cs);
    const double mag = (k ? 2.0 : 1.0) * bessel_j_series(k, zh * zq);
    const cplx<double> e = {cs * mag, sn * mag};
    switch (k & 3) {  // times i^k
        case 0: return e;
        case 1: return {-e.im, e.re};
        case 2: return {-e.re, -e.im};
        default: return {e.im, -e.re};
    }
}

// HERM (Hermitian strengths): the grid holds two transforms per frequency instead of four --
//   T1 = F[c_00 + i c_11]  (both real),   T2 = F[c_01]   (c_10 = conj(c_01)) --
// and the four products are rebuilt from their values at the target s and at its mirror image -s:
//   V_00(s) = (T1(s) + conj(T1(-s))) / 2,   V_11(s) = (T1(s) - conj(T1(-s))) / 2i,
//   V_01(s) = T2(s),                        V_10(s) = conj(T2(-s)),
// exact identities of the non-uniform DFT of real / conjugate-paired strengths.  The caller plans a box
// that is symmetric about s = 0, so that -s is a target like any other.  herm = 2: all four strengths are
// real (real-valued Jones matrices on both sides, unpolarized sky), T2 = F[c_01 + i c_10] and
//   V_01(s) = (T2(s) + conj(T2(-s))) / 2,   V_10(s) = (T2(s) - conj(T2(-s))) / 2i.
// NR = rows of a footprint fetched per batch (9 for w <= 9, else 16): their loads are issued back to back in
// straight-line code -- row indices clamped to the footprint, weights beyond it zero -- so that NR (HERM: 2 NR)
// loads are in flight per lane group.  (With a uniform `if (rr < w)` around each row the compiler put every
// load in a basic block of its own, followed by s_waitcnt vmcnt(0): one load in flight per wave, and the
// gather spent 70 % of its time waiting for them one at a time.)
// ZD (direct third dimension, InterpArgs::zd_n) and WT (height terms, InterpArgs::wt_k) are compile-time: carried as run-time
// branches they cost the plain 2-D gather 47 registers (146 -> 193 fp64, 98 -> 177 fp32: a wave per SIMD, 11-24 % of its time).
template <typename T, int DIM, bool HERM, int NR, bool ZD = false, bool WT = false>
__global__ __launch_bounds__(INTERP_THREADS) void k_interp(
    const cplx<T> *__restrict__ grid, int64_t N, const T *__restrict__ bt0,
    const T *__restrict__ bt1, const T *__restrict__ bt2, const int *__restrict__ bl_idx,
    const signed char *__restrict__ flip, const double *__restrict__ scale, InterpArgs a,
    KerParams ker, cplx<T> *__restrict__ out, const cplx<T> *__restrict__ coef,
    const int *__restrict__ ant1, const int *__restrict__ ant2, const int *__restrict__ ustart,
    const int *__restrict__ upairs) {
    const int tid = threadIdx.x;
    const int g = tid & (GROUP - 1);
    const int lane_base = (tid & 63) & ~(GROUP - 1);
    // XCD x (= block id mod 8) takes the x-th eighth of the (frequency, target) items, a contiguous range: targets
    // next to each other in the caller's list -- redundant baselines, when the list is ordered by baseline vector --
    // then read their common grid lines through ONE L2 instead of up to eight
    constexpr int IPW = INTERP_THREADS / GROUP;
    const int64_t per_xcd = a.items_per_xcd;
    const int64_t item = (int64_t)(blockIdx.x & 7) * per_xcd + (int64_t)(blockIdx.x >> 3) * IPW + tid / GROUP;
    if ((int64_t)(blockIdx.x >> 3) * IPW + tid / GROUP >= per_xcd || item >= N * a.nfg) return;  // whole group exits together
    const int fg = (int)(item / N);
    // Redundant baselines (ustart != nullptr): N counts the DISTINCT target vectors of the caller's list, entries
    // [ustart[i], ustart[i + 1]) of the (u, v)-ordered list share target i -- the same point of the transform -- so it
    // is gathered once, at the first member's coordinates, and the 16 lanes then write every member's output slot
    // (each with its own conjugation / feed transposition / eigenbeam coefficients).
    // Packed runs (HERM) evaluate every target at s AND -s: a run of baselines b and the run of baselines -b are the same
    // two evaluations, so they are one item (upairs[2 i], upairs[2 i + 1] = the two runs, the second may be -1) and the
    // second run takes its values with the roles of the two sides swapped.
    const int64_t ui0 = item % N;
    const int64_t ui = upairs ? upairs[2 * ui0] : ui0;
    const int64_t uj = upairs ? upairs[2 * ui0 + 1] : -1;
    const int64_t m0 = ustart ? ustart[ui] : ui, m1 = ustart ? ustart[ui + 1] : ui + 1;
    const int64_t n0 = uj >= 0 ? ustart[uj] : 0, n1 = uj >= 0 ? ustart[uj + 1] : 0;  // members of the mirror run
    const int64_t kl = m0;
    const int64_t k = bl_idx ? bl_idx[kl] : kl;
    const double sg = ((flip && flip[kl]) != (a.negate_all != 0)) ? -1.0 : 1.0;
    const double sc = scale[fg];
    const int w = a.w;
    const T beta = (T)ker.beta, c4 = (T)ker.c;
    const T *bt[3] = {bt0, bt1, bt2};
    double sv[DIM], th[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        sv[d] = sc * sg * (double)bt[d][k];                     // actual target coordinate
        th[d] = a.h[d] * (sv[d] - sc * a.btc[d]);               // theta = h (s - s_c)
    }
    // direct third dimension (2-D instantiations, see InterpArgs::zd_n): the target's third coordinate
    constexpr bool zd = ZD && DIM == 2;
    double svz = 0.0, thz = 0.0;
    if (zd) {
        svz = sc * sg * (double)bt2[k];
        thz = a.zd_h * (svz - sc * a.zd_btc);
    }
    // psi_1_hat at every theta (an even function: the mirror target shares it): quadrature nodes split
    // over the 16 lanes
    double hh[DIM], hhz = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) hh[d] = 0.0;
    for (int q = g; q < ker.nq; q += GROUP) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) hh[d] += ker.glf[q] * cos(th[d] * ker.glz[q]);
        if (zd) hhz += ker.glf[q] * cos(thz * ker.glz[q]);
    }
#pragma unroll
    for (int off = GROUP / 2; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) hh[d] += __shfl_xor(hh[d], off, 64);
        if (zd) hhz += __shfl_xor(hhz, off, 64);
    }
    double den = 1.0, ph = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        den *= hh[d];
        ph += sv[d] * a.xc[d];  // post-phase exp(i s . x_c)
    }
    if (zd) {
        den *= hhz;
        ph += svz * a.zd_xc;
    }
    double pr = 1.0 / den, pi_ = 0.0;
    if (ph != 0.0) {
        double sn, cs;
        sincos(ph, &sn, &cs);
        pi_ = pr * sn;
        pr = pr * cs;
    }

    const int64_t row_sz = (int64_t)a.P[0] * a.cnt[0];
    const int64_t slab_sz = DIM == 2 && a.ctab ? row_sz * a.ncc : row_sz * a.P[1] * a.cnt[1];
    const int64_t plane_sz = DIM == 3 ? slab_sz * a.P[2] * a.cnt[2] : slab_sz;
    const int nouter = DIM == 3 ? w : 1;
    constexpr int NSIDE = HERM ? 2 : 1;   // the target and (HERM) its mirror image
    constexpr int NVAL = HERM ? 2 : 16;   // transforms per frequency held at once (non-HERM: streamed)
    double vre[NSIDE][HERM ? NVAL : 1], vim[NSIDE][HERM ? NVAL : 1];  // compile-time indexed: registers
    // Both sides' footprints first -- origins, this lane's kernel values, the row offsets (with a column plan: NR table
    // lookups per side) -- so that the two sides' lookups travel together, then the sides' row loads: a wave's time is
    // its chain of dependent round trips (PMC: 65 % of a gather wave's 31 us is spent waiting)
    int j0s[NSIDE][DIM], roffs[NSIDE][NR], gcols[NSIDE];
    T kvs[NSIDE][DIM];
#pragma unroll
    for (int side = 0; side < NSIDE; ++side) {
        const double sgn = side ? -1.0 : 1.0;
        int (&j0)[DIM] = j0s[side];
        T (&kv)[DIM] = kvs[side];  // this lane's kernel value along each dimension (lane = footprint offset)
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            const double e = sgn * th[d] * a.n2[d] * (0.5 / M_PI) + 0.5 * a.no[d];
            int j = (int)ceil(e - 0.5 * w);
            j = max(0, min(a.no[d] - w, j));
            j0[d] = j;
            kv[d] = g < w ? es_eval<T>((T)((double)(j + g) - e), beta, c4) : T(0);
        }
        // row offsets of the footprint in the (residue-major) slow dimension: position of index i is
        // (i mod P) cnt + i / P, walked incrementally from one division per side; rows beyond w repeat the last
        int (&roff)[NR] = roffs[side];
        {
            const int P1 = a.P[1], c1 = a.cnt[1];
            int q1 = j0[1] / P1, r1 = j0[1] - q1 * P1;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                int posr = r1 * c1 + q1;
                if (DIM == 2 && a.ctab) {  // uniform
                    const int t = a.ctab[(int64_t)fg * a.ctab_stride + posr];
                    if (t == 0 && rr < w && g == 0 && a.err) atomicAdd(a.err, 1);
                    posr = max(t - 1, 0);
                }
                roff[rr] = posr * (int)row_sz;
                const int inc = rr + 1 < w ? 1 : 0;
                r1 += inc;
                const int wrap = r1 == P1 ? 1 : 0;
                r1 = wrap ? 0 : r1;
                q1 += wrap;
            }
        }
        gcols[side] = out_pos(min(j0[0] + g, a.no[0] - 1), a.P[0], a.cnt[0]);
    }
#pragma unroll
    for (int side = 0; side < NSIDE; ++side) {
        const double sgn = side ? -1.0 : 1.0;
        const int (&j0)[DIM] = j0s[side];
        const T (&kv)[DIM] = kvs[side];
        const int (&roff)[NR] = roffs[side];
        const int gcol = gcols[side];
        T k1[NR];  // row weights: zero beyond the footprint (kv of lanes >= w is zero)
#pragma unroll
        for (int r = 0; r < NR; ++r) k1[r] = __shfl(kv[1], lane_base + r, 64);
        const double pis = side ? -pi_ : pi_;  // exp(i (-s) . x_c) = conj
        constexpr int RUNROLL = HERM ? NVAL : 1;  // HERM: two transforms, compile-time indexed results
#pragma unroll RUNROLL
        for (int r = 0; r < (HERM ? NVAL : a.tpol); ++r) {
            const cplx<T> *plane = grid + ((int64_t)fg * a.tpol + r) * (zd ? plane_sz * a.zd_n : plane_sz) + gcol;
            T sr = T(0), si = T(0);
            if (zd) {
                // the zd_n planes of this transform, each gathered like a 2-D transform and turned by its phase
                // exp(i (k - zd_n / 2) theta_z) (at the mirror target: -theta_z); the rotation walks from plane to plane
                double er, ei, dr, di;
                sincos(sgn * thz, &di, &dr);
                sincos(-sgn * thz * (double)(a.zd_n / 2), &ei, &er);
                double ar_ = 0.0, ai_ = 0.0;
                for (int zk = 0; zk < a.zd_n; ++zk) {
                    const cplx<T> *slab = plane + (int64_t)zk * plane_sz;
                    cplx<T> v[NR];
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) v[rr] = slab[roff[rr]];
                    T tr = T(0), ti = T(0);
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) {
                        tr += v[rr].re * k1[rr];
                        ti += v[rr].im * k1[rr];
                    }
                    ar_ += (double)tr * er - (double)ti * ei;
                    ai_ += (double)tr * ei + (double)ti * er;
                    const double e2 = er * dr - ei * di;
                    ei = er * di + ei * dr;
                    er = e2;
                }
                sr = (T)ar_;
                si = (T)ai_;
            }
            for (int ro = 0; ro < (zd ? 0 : nouter); ++ro) {
                T k2 = T(1);
                const cplx<T> *slab = plane;
                if (DIM == 3) {
                    k2 = __shfl(kv[DIM - 1], lane_base + ro, 64);
                    slab += (int64_t)out_pos(j0[DIM - 1] + ro, a.P[DIM - 1], a.cnt[DIM - 1]) * slab_sz;
                }
                cplx<T> v[NR];
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) v[rr] = slab[roff[rr]];
                T tr = T(0), ti = T(0);
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) {
                    tr += v[rr].re * k1[rr];
                    ti += v[rr].im * k1[rr];
                }
                sr += tr * k2;
                si += ti * k2;
            }
            sr *= kv[0];
            si *= kv[0];
#pragma unroll
            for (int off = GROUP / 2; off > 0; off >>= 1) {
                sr += __shfl_xor(sr, off, 64);
                si += __shfl_xor(si, off, 64);
            }
            const double vr0 = (double)sr * pr - (double)si * pis;
            const double vi0 = (double)sr * pis + (double)si * pr;
            if constexpr (HERM) {
                vre[side][r] = vr0;
                vim[side][r] = vi0;
                continue;
            }
            for (int64_t m = m0 + g; m < m1; m += GROUP) {  // the target's members, dealt over the 16 lanes (no list: lane 0)
                const int64_t km = bl_idx ? bl_idx[m] : m;
                const bool neg = (flip && flip[m]) != (a.negate_all != 0);
                double vr = vr0, vi = vi0;
                if constexpr (WT) {  // this member's height factor, before the conjugation
                    const cplx<double> f = wterm_factor(a.wt_k, a.wt_zc, a.wt_zh, sc * (neg ? -1.0 : 1.0) * (double)((const T *)a.wt_bz)[km]);
                    vr = vr0 * f.re - vi0 * f.im;
                    vi = vr0 * f.im + vi0 * f.re;
                }
                const double vim_ = neg ? -vi : vi;  // conj for flipped baselines (cpu_simulate.py:298)
                const int rt = a.transpose_flipped && neg && a.tpol == 4 ? (r & 1) * 2 + (r >> 1) : r;
                const int64_t po = rt < 16 ? a.out_pol_off[rt] : (int64_t)rt * a.out_pol_off[1];
                cplx<T> *ob = out + (int64_t)fg * a.out_fg_stride + km * a.out_k_stride;
                cplx<T> *o = ob + po;
                if (a.basis) {
                    const int f = a.f_first + fg;
                    const int64_t cs1 = (int64_t)ant1[km] * a.nbasis, cs2 = (int64_t)ant2[km] * a.nbasis;
                    const cplx<T> c1k = coef[(cs1 + a.kk) * a.ncoef_freq + f];
                    const cplx<T> c2l = coef[(cs2 + a.ll) * a.ncoef_freq + f];
                    const cplx<double> w1 = cmul(cplx<double>{(double)c1k.re, -(double)c1k.im},
                                                 cplx<double>{(double)c2l.re, (double)c2l.im});
                    const cplx<double> v1 = cmul(w1, cplx<double>{vr, vim_});
                    if (a.basis_part != 2) {
                        o->re += (T)v1.re;
                        o->im += (T)v1.im;
                    }
                    if (a.kk != a.ll && a.basis_part != 1) {
                        const cplx<T> c1l = coef[(cs1 + a.ll) * a.ncoef_freq + f];
                        const cplx<T> c2k = coef[(cs2 + a.kk) * a.ncoef_freq + f];
                        const cplx<double> w2 = cmul(cplx<double>{(double)c1l.re, -(double)c1l.im},
                                                     cplx<double>{(double)c2k.re, (double)c2k.im});
                        const cplx<double> v2 = cmul(w2, cplx<double>{vr, vim_});
                        const int rs = (r & 1) * 2 + (r >> 1);  // feed-transposed slot (V.swapaxes(1, 2))
                        cplx<T> *o2 = ob + a.out_pol_off[rs];
                        o2->re += (T)v2.re;
                        o2->im += (T)v2.im;
                    }
                } else if (a.accumulate) {
                    o->re += (T)vr;
                    o->im += (T)vim_;
                } else {
                    *o = {(T)vr, (T)vim_};
                }
            }
        }
    }
    if constexpr (HERM) {
        // P = T1(s), M = T1(-s), C = T2(s), D = T2(-s)
      for (int sec = 0; sec < 2; ++sec) {  // the run of the target, then (packed pairs of runs) the run of its mirror image,
        if (sec && n1 <= n0) break;       // which sees the two sides the other way round (uniform over the 16 lanes)
        const double Pr = sec ? vre[1][0] : vre[0][0], Pi = sec ? vim[1][0] : vim[0][0], Mr = sec ? vre[0][0] : vre[1][0],
                     Mi = sec ? vim[0][0] : vim[1][0];
        double o_re[4], o_im[4];
        o_re[0] = 0.5 * (Pr + Mr);   // (P + conj M) / 2
        o_im[0] = 0.5 * (Pi - Mi);
        o_re[3] = 0.5 * (Pi + Mi);   // (P - conj M) / 2i = -i/2 ((Pr - Mr) + i (Pi + Mi))
        o_im[3] = -0.5 * (Pr - Mr);
        const double Cr = sec ? vre[1][1] : vre[0][1], Ci = sec ? vim[1][1] : vim[0][1], Dr = sec ? vre[0][1] : vre[1][1],
                     Di = sec ? vim[0][1] : vim[1][1];
        if (a.herm == 2) {  // all four strengths real: T2 = F[c_01 + i c_10], unpacked like T1
            o_re[1] = 0.5 * (Cr + Dr);
            o_im[1] = 0.5 * (Ci - Di);
            o_re[2] = 0.5 * (Ci + Di);
            o_im[2] = -0.5 * (Cr - Dr);
        } else {            // Hermitian strengths: T2 = F[c_01], c_10 = conj c_01
            o_re[1] = Cr;   // C
            o_im[1] = Ci;
            o_re[2] = Dr;   // conj D
            o_im[2] = -Di;
        }
        for (int64_t m = (sec ? n0 : m0) + g; m < (sec ? n1 : m1); m += GROUP) {  // the run's members, dealt over the 16 lanes (no list: lane 0)
            const int64_t km = bl_idx ? bl_idx[m] : m;
            const bool neg = (flip && flip[m]) != (a.negate_all != 0);
            cplx<T> *ob = out + (int64_t)fg * a.out_fg_stride + km * a.out_k_stride;
            cplx<double> w1 = {1.0, 0.0}, w2 = {0.0, 0.0};
            if (a.basis) {  // eigenbeam term (k, l): vis += conj(C[a1,k]) C[a2,l] V  (+ the transposed (l, k) term)
                const int f = a.f_first + fg;
                const int64_t cs1 = (int64_t)ant1[km] * a.nbasis, cs2 = (int64_t)ant2[km] * a.nbasis;
                const cplx<T> c1k = coef[(cs1 + a.kk) * a.ncoef_freq + f];
                const cplx<T> c2l = coef[(cs2 + a.ll) * a.ncoef_freq + f];
                w1 = cmul(cplx<double>{(double)c1k.re, -(double)c1k.im}, cplx<double>{(double)c2l.re, (double)c2l.im});
                if (a.kk != a.ll) {  // cpu_simulate.py:464-468
                    const cplx<T> c1l = coef[(cs1 + a.ll) * a.ncoef_freq + f];
                    const cplx<T> c2k = coef[(cs2 + a.kk) * a.ncoef_freq + f];
                    w2 = cmul(cplx<double>{(double)c1l.re, -(double)c1l.im}, cplx<double>{(double)c2k.re, (double)c2k.im});
                }
            }
            cplx<double> wf = {1.0, 0.0};
            if constexpr (WT)  // this member's height factor, applied before the conjugation
                wf = wterm_factor(a.wt_k, a.wt_zc, a.wt_zh, sc * (neg ? -1.0 : 1.0) * (double)((const T *)a.wt_bz)[km]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double xr = o_re[r] * wf.re - o_im[r] * wf.im, xi = o_re[r] * wf.im + o_im[r] * wf.re;
                const double vr = xr, vi = neg ? -xi : xi;  // conj for flipped baselines
                cplx<T> *o = ob + a.out_pol_off[a.transpose_flipped && neg ? (r & 1) * 2 + (r >> 1) : r];
                if (a.basis) {
                    const cplx<double> v1 = cmul(w1, cplx<double>{vr, vi});
                    o->re += (T)v1.re;
                    o->im += (T)v1.im;
                    if (a.kk != a.ll) {
                        const cplx<double> v2 = cmul(w2, cplx<double>{vr, vi});
                        cplx<T> *o2 = ob + a.out_pol_off[(r & 1) * 2 + (r >> 1)];  // feed-transposed slot
                        o2->re += (T)v2.re;
                        o2->im += (T)v2.im;
                    }
                } else if (a.accumulate) {
                    o->re += (T)vr;
                    o->im += (T)vi;
                } else {
                    *o = {(T)vr, (T)vi};
                }
            }
        }
      }
    }
}

// Output mask of a column plan's y-pass (RowDifArgs::omask): every distinct target (and its mirror image where the
// run gathers there too) marks, for the compact column blocks its footprint columns fall in, the 16-output chunks of
// each residue that hold one of its footprint rows.  Footprints are placed as k_interp places them; one whose first
// cell is within 1e-6 of a rounding boundary marks a cell more on either side.
struct RowMaskArgs {
    int w, nfg, sides;
    int n2x, nox, Px, cntx, n2y, noy, Py, Qy;
    double hx, btcx, hy, btcy;
    int blk_log, nblk, nw, ctab_stride;
};
template <typename T>
__global__ void k_plan_rowmask(int64_t NU, const T *__restrict__ btx, const T *__restrict__ bty,
                               const int *__restrict__ bl_idx, const signed char *__restrict__ flip,
                               const int *__restrict__ ustart, const double *__restrict__ scale, RowMaskArgs a,
                               const int *__restrict__ ctab, unsigned long long *__restrict__ mask) {
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= NU * a.nfg * a.sides) return;
    const int side = (int)(item % a.sides);
    const int64_t rest = item / a.sides;
    const int fg = (int)(rest / NU);
    const int64_t ui = rest % NU;
    const int64_t kl = ustart ? ustart[ui] : ui;
    const int64_t k = bl_idx ? bl_idx[kl] : kl;
    const double sg = (flip && flip[kl]) ? -1.0 : 1.0, sgn = side ? -1.0 : 1.0, sc = scale[fg];
    int lo[2], hi[2];
    const double hh[2] = {a.hx, a.hy}, bc[2] = {a.btcx, a.btcy};
    const int n2[2] = {a.n2x, a.n2y}, no[2] = {a.nox, a.noy};
    const double b[2] = {(double)btx[k], (double)bty[k]};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const double sv = sc * sg * b[d];
        const double th = hh[d] * (sv - sc * bc[d]);
        const double e = sgn * th * n2[d] * (0.5 / M_PI) + 0.5 * no[d];
        const double t = e - 0.5 * a.w, jc = ceil(t);
        const bool amb = jc - t < 1e-6 || jc - t > 1.0 - 1e-6;
        lo[d] = max(0, min(no[d] - a.w, (int)jc));
        hi[d] = lo[d] + a.w - 1;
        if (amb) {
            lo[d] = max(0, lo[d] - 1);
            hi[d] = min(no[d] - 1, hi[d] + 1);
        }
    }
    const int half = a.noy / 2;
    for (int cx = lo[0]; cx <= hi[0]; ++cx) {
        const int t = ctab[(int64_t)fg * a.ctab_stride + out_pos(cx, a.Px, a.cntx)];
        if (!t) continue;  // (only a cell added for rounding here and not by the host's plan)
        const int64_t blk = (t - 1) >> a.blk_log;
        for (int cy = lo[1]; cy <= hi[1]; ++cy) {
            const int l = cy - half;
            int pp = l % a.Py;
            if (pp < 0) pp += a.Py;
            const int ks = (l - pp) / a.Py;
            const int c = (ks < 0 ? ks + a.Qy : ks) >> 4;
            atomicOr(&mask[(((int64_t)fg * a.nblk + blk) * a.Py + pp) * a.nw + (c >> 6)], 1ull << (c & 63));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host-side plan
// ---------------------------------------------------------------------------------------------
// One (k, l) term of the eigenbeam contraction, handed to Nufft3::interp.
struct BasisTerm {
    const void *coef;   // device (nant, K, nfreq) complex
    const int *ant1, *ant2;  // device (nbls) antenna index of each baseline
    int kk, ll, nbasis, nfreq, f_first;
    int part = 0, negate = 0;  // InterpArgs::basis_part / negate_all
};

// Height term handed to Nufft3::interp (InterpArgs::wt_*)
struct WTerm {
    int k;
    double zc, zh;
    const void *bz;
};

inline bool rowfft_uses_st(const DimGeom &g, bool col);
inline void rowfft_shape(const DimGeom &g, bool col, int &tpr, int &rpw);

template <typename T>
class Nufft3 {
   public:
    int dim;
    bool transpose_flipped = false;  // InterpArgs::transpose_flipped / FusedArgs::tflip (fv_sim_set_reference_compat)
    double eps, sigma;
    KerParams ker;
    Geom geo;
    hipStream_t stream;

    // ---- fused gather plan (see FgHdr): built once per (geometry, target set), reused by every fft()
    DevBuf fg_recs, fg_meta, fg_start, fg_list;
    struct FusedKey {
        int64_t serial = -1, N = 0, ofs = 0, oks = 0, targets = 0;  // targets: caller's version of the device arrays' contents
        const void *btx = nullptr, *bl_idx = nullptr, *flip = nullptr, *scale = nullptr;
        int nfg = 0, tpol = 0;
        double g[10] = {0};  // h, btc, xc, no, n2 of both dimensions
        bool operator==(const FusedKey &o) const {
            for (int i = 0; i < 10; ++i)
                if (g[i] != o.g[i]) return false;
            return serial == o.serial && targets == o.targets && N == o.N && ofs == o.ofs && oks == o.oks && btx == o.btx &&
                   bl_idx == o.bl_idx && flip == o.flip && scale == o.scale && nfg == o.nfg && tpol == o.tpol;
        }
    } fused_key;
    FusedArgs fused_args{};
    bool fused_active = false, last_fft_fused = false;
    // Column plan (2-D, blocked B, stand-alone gather): the targets of a regular array read only a fraction of the
    // transform's columns in the first dimension (HERA-350: a third -- its baselines sit on a lattice); the x-pass
    // then stores ONLY those, compacted, the y-pass transforms only those, and the gather finds them through the
    // table.  col_tab[fg * x.nos() + position] = 1 + compact column | 0 (device; built by the caller from its targets,
    // per frequency of the group: tpol transforms share an entry), col_ncc = compact columns (largest over the
    // frequencies).  Armed by the caller before fft(); nullptr = every column.
    const int *col_tab = nullptr;   // the gather's view: 1 + compact column
    const int *col_xtab = nullptr;  // the x-pass's view: 1 + element index inside a row's blocked output (RowDifArgs::ctab)
    int col_tab_tpol = 1, col_ncc = 0;
    const int *first_pass_ext = nullptr;  // RowDifArgs::row_ext of the rowfft call in flight (fft())
    double grid_slack = -1.0;
    // the slack share a geometry takes: the caller's, else the default on HBM-bound grids (>= 4e6 fine-grid cells) and
    // none on small ones (latency-bound: C2 runs 2 % faster on the tight grid)
    double slack_for(const double *X, const double *B, double scale_max) const {
        if (grid_slack >= 0.0) return grid_slack;
        double cells = 1.0;
        for (int d = 0; d < dim; ++d) {
            DimGeom g;
            g.X = X[d];
            g.B = B[d];
            set_dim_geom(g, sigma, ker.w, scale_max, true, 0.0);
            cells *= g.n2;
        }
        return cells >= 4.0e6 ? -1.0 : 0.0;
    }  // set_dim_geom's slack_share for the next set_geometry / plan_buffer_cells (-1: the default)
    int *col_err = nullptr;
    const unsigned long long *col_omask = nullptr;  // RowDifArgs::omask of the y-pass (k_plan_rowmask), or nullptr
    int col_omask_nblk = 0;
    double col_out_cells = 0;  // cells of C per transform the masked y-pass stores (0: all)
    void arm_columns(const int *tab, const int *xtab, int tpol, int ncc, int *err = nullptr, const unsigned long long *omask = nullptr,
                     int omask_nblk = 0) {
        col_tab = tab && xtab && (dim == 2 || zdirect) && b_block_log() ? tab : nullptr;
        col_xtab = col_tab ? xtab : nullptr;
        col_tab_tpol = tpol;
        col_ncc = ncc;
        col_err = err;
        col_omask = col_tab ? omask : nullptr;
        col_omask_nblk = omask_nblk;
        col_out_cells = 0;
    }
    int ypass_cols_log() const {  // log2 of the columns a y-pass workgroup owns (column mode)
        int tpr, rpw;
        rowfft_shape(geo.d[1], true, tpr, rpw);
        return ilog2_c(rpw);
    }
    // builds the y-pass output mask of a column plan from the plan's targets (device arrays) on `stream`
    void build_rowmask(int64_t NU, const T *btx, const T *bty, const int *bl_idx, const signed char *flip, const int *ustart,
                       const double *scale_dev, int nfg, bool both, const int *ctab, int ncc, unsigned long long *mask,
                       int nblk, int nw) const {
        const DimGeom &x = geo.d[0], &y = geo.d[1];
        RowMaskArgs a{};
        a.w = ker.w;
        a.nfg = nfg;
        a.sides = both ? 2 : 1;
        a.n2x = x.n2; a.nox = x.no; a.Px = x.sP(); a.cntx = x.cnt(); a.hx = x.h; a.btcx = x.btc;
        a.n2y = y.n2; a.noy = y.no; a.Py = y.P; a.Qy = y.Q; a.hy = y.h; a.btcy = y.btc;
        a.blk_log = ypass_cols_log();
        a.nblk = nblk;
        a.nw = nw;
        a.ctab_stride = x.nos();
        (void)ncc;
        const int64_t items = NU * nfg * a.sides;
        if (items == 0) return;
        hipLaunchKernelGGL(k_plan_rowmask<T>, dim3((unsigned)cdiv(items, 256)), dim3(256), 0, stream, NU, btx, bty, bl_idx,
                           flip, ustart, scale_dev, a, ctab, mask);
    }
    int xcols() const { return col_tab ? col_ncc : geo.d[0].nos(); }  // columns of B / rows of C per transform
    int b_block_log_public() const { return b_block_log(); }
    // (3-D with the direct third dimension: every (transform, z) plane is a 2-D transform read by the same targets)
    bool columns_possible() const { return (dim == 2 || zdirect) && b_block_log() != 0 && y_reads_columns() && !fused_possible(); }
    bool fused_possible() const;
    // Arms the fused gather for the next fft() (which then leaves no grid for interp()); false when the
    // configuration does not qualify and the caller must use interp().
    bool prepare_fused_gather(int64_t N, const T *btx, const T *bty, const int *bl_idx, const signed char *flip,
                              const double *scale_dev, int nfg, int tpol, cplx<T> *out, int64_t out_fg_stride,
                              int64_t out_k_stride, const int64_t *out_pol_off, int64_t targets_serial = 0,
                              cplx<T> *out_mate = nullptr);
    int64_t M = 0;            // sources currently binned
    int64_t geom_serial = 0;  // bumps whenever the source->cell mapping changes

    // device state
    DevBuf i0u, fu, tile_of, binmeta, bin_start, i0s, fs, perm, slot_id, kw, scan_tot, scan_off, scan_tot2, scan_off2;
    const int *Mp = nullptr;   // device-side live source count (optional)
    int *oob_ptr = nullptr;
    int *err_oob = nullptr;    // owner's sticky counter of clamped / NaN sources (else the plan's own, reset per sort)
    DevBuf dec[3], tw[3];  // (set_fft_geometry's twiddles; set_geometry keeps its tables in tab_cache)
    // Deconvolution / twiddle tables of a dimension depend on (na, n2) only and a run cycles through the same two dozen
    // geometries every time step: built once each and kept (they were 1.3 % of a C3 step, rebuilt 46 times per time
    // step), and a table is never rewritten while an earlier launch may still read it.
    struct DimTables {
        DevBuf dec, tw;
    };
    std::map<std::pair<int, int>, std::unique_ptr<DimTables>> tab_cache[3];
    const T *dec_cur[3] = {nullptr, nullptr, nullptr};
    const cplx<T> *tw_cur[3] = {nullptr, nullptr, nullptr};
    DevBuf buf0, buf1;  // ping-pong: A -> (x-pass) B -> (transpose) Bt -> (y-pass) Ct
    DevBuf strengths;   // [M][ntrans] sorted order

    // Direct third dimension (3-D): the spread grid is na_z planes along z -- for arrays that are anywhere near flat
    // nearly all of them kernel width, not source range -- and the targets are few (10^4 - 10^5) against the 10^7
    // (x, y) columns of the transform.  A z-pass would transform EVERY column (read na_z planes, write no_z: at C3z
    // 14 GB per transform, two thirds of the whole 3-D FFT's time) for the gather to read a w_z-wide sliver of a
    // thousandth of them; instead the x- and y-passes run per (transform, z) plane exactly as in 2-D and every target
    // sums the na_z planes itself with their exact phases (InterpArgs::zd_n): na_z / w_z times the gather's loads, no
    // z-pass, no inner kernel along z (one approximation less), and na_z no longer rounded up to whole bins.
    // FFTVIS_HIP_NO_ZDIRECT=1: the three-pass transform.
    bool zdirect = false;
    Nufft3(int dim_, double eps_, double sigma_, hipStream_t s, int w_override = 0)
        : dim(dim_), eps(eps_), sigma(sigma_), stream(s) {
        zdirect = dim_ == 3 && !std::getenv("FFTVIS_HIP_NO_ZDIRECT");
        FV_REQUIRE(dim == 2 || dim == 3, "dim must be 2 or 3");
        FV_REQUIRE(sigma == 2.0 || sigma == 1.25, "upsample factor must be 2 or 1.25");
        FV_REQUIRE(eps > 0 && eps < 1, "eps must be in (0, 1)");
        ker = make_kernel(eps, sigma, w_override);
        // 3-D transforms carry a larger error constant (seeded fuzzing: 12 eps on band-edge baselines of a
        // non-coplanar array at every eps, against <= 2 eps in 2-D): one more cell of kernel width there
        if (dim == 3 && !w_override && ker.w < (sigma == 2.0 ? MAX_W : 15)) ker = make_kernel(eps, sigma, ker.w + 1);
        // fp32 at sigma = 1.25: beyond ten cells a wider kernel only amplifies rounding (the kernel's transform falls by
        // ~e^{-w/2} per dimension across the band and fp32 has 7 digits to lose): measured rel. l2 against exact sums,
        // eps asked 1e-4 / 2e-5 / 1e-6 / 6e-8 / 1e-9 -> 3.5e-6 / 3.1e-6 / 6.7e-6 / 4.6e-5 / 3.6e-4 (2-D; w = 8 .. 15) and
        // 3.4e-6 / 6.7e-6 / 5.4e-5 / 5.7e-4 / 2.0e-3 (3-D) -- and 6e-8 is the fp32 DEFAULT tolerance.  sigma = 2 is flat at
        // 2.8e-6 below eps = 1e-5 and needs no cap.  (finufft clamps eps to the type's epsilon, not the width.)
        // Three dimensions lose e^{-w/2} three times: targets in the corners of the box sit at ~e^{-3w/2} of the centre --
        // a non-coplanar array with 4.7 m of height range (fuzz seed 7000 / 346): w = 7 / 9 / 10 -> 2.4e-4 / 3.5e-4 /
        // 1.8e-3; capped at 8 there (the sigma = 1.25 floor of fp32 in 3-D is the 1e-3 the automatic choice assumes).
        const int w_cap32 = dim == 3 ? 8 : 10;
        if (sizeof(T) == 4 && sigma != 2.0 && !w_override && ker.w > w_cap32) ker = make_kernel(eps, sigma, w_cap32);
        geo.dim = dim;
    }

    // Launch order of the spread workgroups (groups of 4 blocks of one bin row), heaviest first.
    // A wave's time grows with the sources that reach its block, and skies are far from uniform on
    // the grid: directions uniform on the sphere pile up towards the rim of the (l, m) disc like
    // 1 / sqrt(1 - r^2) (7x the central density in the outermost blocks of C2).  Dispatched in
    // raster order, the heavy rim rows come last and the kernel ends in a long, nearly empty
    // tail (measured: 4.8 resident waves per CU on average); heaviest-first lets the light central
    // blocks fill that tail.  The weight is only a heuristic -- any order is correct.
    // One table per (nbx, nby), kept for the life of the plan: frequency groups cycle through a
    // handful of grid sizes every time step.
    // (Shared between the plans of one simulator -- its lanes cycle through the same geometries -- and filled by a
    // blocking copy that does not involve the plans' streams: a cold handle met 23 grid sizes x 4 lanes in its first
    // two time steps, each a weight table of 10^5 entries, a sort and a stream synchronisation: 0.4 s of a 2 s C3 call.)
    // Disc (disc_radius > 0, 2-D): the caller's sources are projections of unit vectors, x^2 + y^2 <= disc_radius^2 in the
    // source coordinates whatever the box -- a fifth of a square grid's blocks can then never be touched.  Those blocks
    // leave the launch list (nobody writes them), and the x-pass reads every row only over the extent the disc allows:
    // row_ext[2 by], row_ext[2 by + 1] = first / one-past-last cell of the 8 rows of block row by, in whole 4-block
    // groups; k_bin_count counts a source outside the disc as out of the box (the run fails).
    struct OrderEntry {
        DevBuf buf;  // n_groups launch entries, then (disc) 2 nby extents
        int n_groups = 0;
        bool ext = false;
        int64_t cells = 0;  // cells of A the launch list writes = cells the x-pass reads
    };
    using OrderKey = std::array<double, 9>;
    using OrderCache = std::map<OrderKey, std::unique_ptr<OrderEntry>>;
    std::shared_ptr<OrderCache> order_cache = std::make_shared<OrderCache>();
    const int *order_ptr = nullptr, *row_ext_ptr = nullptr;
    int order_n = 0;
    int64_t order_cells = 0;  // cells of A per transform that the spread writes and the first pass reads (2-D)
    double disc_radius = 0.0;
    // rounding of the callers' unit vectors: fp64 runs are also handed vectors that were computed in float32 (a coord_mgr
    // or catalog_device of single-precision arrays: norms off by 6e-8), so the fp64 margin is 1e-6 too -- 0.002 cells
    // at the rim of the largest grids, not a block more
    static constexpr double disc_margin() { return sizeof(T) == 4 ? 1e-5 : 1e-6; }
    void build_block_order() {
        const int nbx = geo.nbin[0], nby = geo.nbin[1], ngx = (int)cdiv(nbx, 4);
        // (3-D with the direct third dimension: the (x, y) projections of the sources lie in the same disc on every z-plane)
        const bool disc = disc_radius > 0.0 && (dim == 2 || zdirect) && rowfft_uses_st(geo.d[0], false);  // (the LDS kernel reads whole rows)
        OrderKey key{(double)nbx, (double)nby, 0, 0, 0, 0, 0, 0, 0};
        if (disc) key = {(double)nbx, (double)nby, geo.d[0].xc, geo.d[1].xc, geo.d[0].h, geo.d[1].h, (double)geo.d[0].na, (double)ker.w, disc_radius};
        auto hit = order_cache->find(key);
        if (hit != order_cache->end()) {
            order_ptr = hit->second->buf.template as<int>();
            order_n = hit->second->n_groups;
            order_cells = hit->second->cells;
            row_ext_ptr = hit->second->ext ? order_ptr + order_n : nullptr;
            return;
        }
        // extents of the block rows (cells; all of the row without a disc)
        std::vector<int> ext(2 * (size_t)nby);
        for (int by = 0; by < nby; ++by) {
            int x0 = 0, x1 = geo.d[0].na;
            if (disc) {
                const int w = ker.w;
                const DimGeom &gx = geo.d[0], &gy = geo.d[1];
                // a source at grid position p (cells) touches the cells in [p - w/2, p + w/2]: the block row's cells
                // 8 by .. 8 by + 7 are reached from p in (8 by - w/2, 8 by + 7 + w/2]; one more cell either side for rounding
                const double pl = 8.0 * by - 0.5 * w - 1.0, ph = 8.0 * by + 7.0 + 0.5 * w + 1.0;
                const double yl = (pl - 0.5 * gy.na) * gy.h + gy.xc, yh = (ph - 0.5 * gy.na) * gy.h + gy.xc;
                const double ymin = yl > 0 ? yl : yh < 0 ? -yh : 0.0;  // smallest |y| in the interval
                const double R = disc_radius * (1.0 + disc_margin());
                if (ymin >= R) {
                    x0 = x1 = 0;
                } else {
                    const double xm = std::sqrt(R * R - ymin * ymin);
                    const double pxl = (-xm - gx.xc) / gx.h + 0.5 * gx.na, pxh = (xm - gx.xc) / gx.h + 0.5 * gx.na;
                    const int c0 = (int)std::floor(pxl - 0.5 * w) - 1, c1 = (int)std::ceil(pxh + 0.5 * w) + 2;
                    x0 = std::max(0, c0) / 32 * 32;
                    x1 = std::min(gx.na, (std::max(c1, 0) + 31) / 32 * 32);
                    if (x1 <= x0) x0 = x1 = 0;
                }
            }
            ext[2 * (size_t)by] = x0;
            ext[2 * (size_t)by + 1] = x1;
        }
        auto kept = [&](int by, int gx) { return 32 * gx < ext[2 * (size_t)by + 1] && 32 * gx + 32 > ext[2 * (size_t)by]; };
        std::vector<int> order_host;
        order_host.reserve((size_t)ngx * nby + 2 * (size_t)nby);
        if ((size_t)ngx * nby > 16384) {
            // huge grids are HBM-bound and have thousands of groups per CU: raster order keeps their
            // writes local and the tail is negligible there
            for (int by = 0; by < nby; ++by)
                for (int gx = 0; gx < ngx; ++gx)
                    if (kept(by, gx)) order_host.push_back((by << 16) | gx);
        } else {
            std::vector<std::pair<float, int>> wg;
            for (int by = 0; by < nby; ++by) {
                const double ry = ((by + 0.5) * (1 << BINLOG) - 0.5 * geo.d[1].na) / (0.5 * geo.d[1].na);
                for (int gx = 0; gx < ngx; ++gx) {
                    if (!kept(by, gx)) continue;
                    double wsum = 0;
                    for (int k = 0; k < 4; ++k) {
                        const int bx = gx * 4 + k;
                        if (bx >= nbx) break;
                        const double rx = ((bx + 0.5) * (1 << BINLOG) - 0.5 * geo.d[0].na) / (0.5 * geo.d[0].na);
                        const double r2 = rx * rx + ry * ry;
                        wsum += r2 < 1.0 ? 1.0 / std::sqrt(std::max(1.0 - r2, 0.02)) : 0.05;
                    }
                    wg.push_back({(float)wsum, (by << 16) | gx});
                }
            }
            std::stable_sort(wg.begin(), wg.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
            for (size_t i = 0; i < wg.size(); ++i) order_host.push_back(wg[i].second);
        }
        std::unique_ptr<OrderEntry> e(new OrderEntry);
        e->n_groups = (int)order_host.size();
        e->ext = disc;
        for (int og : order_host) e->cells += (int64_t)std::min(4, nbx - (og & 0xffff) * 4) * 64;
        order_host.insert(order_host.end(), ext.begin(), ext.end());
        e->buf.reserve(sizeof(int) * order_host.size());
        FV_HIP(hipMemcpy(e->buf.p, order_host.data(), sizeof(int) * order_host.size(), hipMemcpyHostToDevice));
        order_ptr = e->buf.template as<int>();
        order_n = e->n_groups;
        order_cells = e->cells;
        row_ext_ptr = disc ? order_ptr + order_n : nullptr;
        (*order_cache)[key] = std::move(e);
    }

    // Upper bound (cells per transform) of the grid buffers a geometry will need: the same grid sizing as
    // set_geometry, host arithmetic only.  Lets a run size its buffers once, before any kernel is queued -- growing
    // them group by group means a hipFree + hipMalloc of gigabytes, each a device synchronisation, in the middle of
    // the first time step (0.45 s of a cold 2 s C3 call).
    int64_t plan_buffer_cells(const double *X, const double *B, double scale_max, int *na_max = nullptr,
                              int *n2_max = nullptr) const {
        DimGeom g[3];
        const double slack = slack_for(X, B, scale_max);
        for (int d = 0; d < dim; ++d) {
            g[d].X = X[d];
            g[d].B = B[d];
            const bool last = zdirect ? d == 1 : d == dim - 1;  // the dimension the gather reads contiguously
            set_dim_geom(g[d], sigma, ker.w, scale_max, last || d == 2, slack);
            if (d > 0) cap_column_q(g[d]);
            g[d].rm = !last && !debug_switch_natural_order();
            if (zdirect && d == 2) g[d].na = g[d].n1;  // planes, not bins
            if (na_max) na_max[d] = std::max(na_max[d], g[d].na);
            if (n2_max) n2_max[d] = std::max(n2_max[d], g[d].n2);
        }
        const int64_t zin = dim > 2 ? g[2].na : 1, zout = dim > 2 ? (zdirect ? g[2].na : g[2].nos()) : 1;
        const int64_t pitch = (g[0].nos() + 7) / 8 * 8;
        return std::max({zin * g[1].na * g[0].na, zin * g[1].na * pitch, zin * pitch * g[1].nos(), zout * pitch * g[1].nos()});
    }
    // grid buffers of `bytes_cells` cells in all, and the per-dimension deconvolution / twiddle tables
    void reserve_buffers(int64_t cells, const int *na_max, const int *n2_max) {
        buf0.reserve(sizeof(cplx<T>) * (size_t)cells);
        buf1.reserve(sizeof(cplx<T>) * (size_t)cells);
        (void)na_max;
        (void)n2_max;
    }

    // Bounds -> grid sizes, deconvolution + twiddle tables.
    void set_geometry(const double *xc, const double *X, const double *btc, const double *B,
                      double scale_max) {
        Geom old = geo;
        const bool first = geom_serial == 0;
        const double slack = slack_for(X, B, scale_max);
        for (int d = 0; d < dim; ++d) {
            geo.d[d].xc = xc[d];
            geo.d[d].X = X[d];
            geo.d[d].btc = btc[d];
            geo.d[d].B = B[d];
            const bool last = zdirect ? d == 1 : d == dim - 1;  // the dimension the gather reads contiguously
            set_dim_geom(geo.d[d], sigma, ker.w, scale_max, last || d == 2, slack);
            if (d > 0) cap_column_q(geo.d[d]);
            // (residue-major storage of the last dimension as well: C3's FFT passes -2.5 %, its gather +41 %)
            geo.d[d].rm = !last && !debug_switch_natural_order();
            geo.nbin[d] = geo.d[d].na >> BINLOG;
            if (zdirect && d == 2) {  // planes the sources can touch (even), in bins of eight, the last one short
                geo.d[d].na = geo.d[d].n1;
                geo.nbin[d] = (int)cdiv(geo.d[d].na, 1 << BINLOG);
            }
        for (int d = dim; d < 3; ++d) geo.nbin[d] = 1;
        }
        if (first || old.nbin[0] != geo.nbin[0] || old.nbin[1] != geo.nbin[1] || disc_radius > 0.0) build_block_order();
        for (int d = 0; d < dim; ++d) {
            const DimGeom &g = geo.d[d];
            if (zdirect && d == 2) {  // no transform along z: neither an inner kernel to undo nor twiddles
                dec_cur[d] = nullptr;
                tw_cur[d] = nullptr;
                continue;
            }
            if (old.d[d].na == g.na && old.d[d].n2 == g.n2 && dec_cur[d] && !first) continue;
            std::unique_ptr<DimTables> &e = tab_cache[d][{g.na, g.n2}];
            if (!e) {
                if (tab_cache[d].size() > 256) {  // (a long-lived plan that met hundreds of geometries: start over)
                    FV_HIP(hipStreamSynchronize(stream));
                    tab_cache[d].clear();
                }
                std::unique_ptr<DimTables> &f = tab_cache[d][{g.na, g.n2}];
                f.reset(new DimTables);
                f->dec.reserve(sizeof(T) * g.na);
                hipLaunchKernelGGL(k_deconv_table<T>, dim3(cdiv(g.na, 256)), dim3(256), 0, stream, g.na,
                                   g.n2, ker, f->dec.template as<T>());
                f->tw.reserve(sizeof(cplx<T>) * g.n2);
                hipLaunchKernelGGL(k_twiddle_table<T>, dim3(cdiv(g.n2, 256)), dim3(256), 0, stream,
                                   g.n2, f->tw.template as<cplx<T>>());
                dec_cur[d] = f->dec.template as<T>();
                tw_cur[d] = f->tw.template as<cplx<T>>();
                continue;
            }
            dec_cur[d] = e->dec.template as<T>();
            tw_cur[d] = e->tw.template as<cplx<T>>();
        }
        bool changed = first;
        for (int d = 0; d < dim; ++d)
            if (old.d[d].xc != geo.d[d].xc || old.d[d].h != geo.d[d].h || old.d[d].na != geo.d[d].na)
                changed = true;
        if (changed) {
            ++geom_serial;
            M = 0;
        }
    }

    // Bin-sort the sources for the current geometry (device pointers, length M each) and
    // tabulate their kernel weights.
    // M_ is the live count, or -- when Mdev is given -- the capacity, with the live count read
    // from device memory by the kernels (no host round trip).
    void set_sources(int64_t M_, const T *x, const T *y, const T *z, const int *Mdev = nullptr) {
        M = M_;
        Mp = Mdev;
        const int nb = geo.nbins();
        const int64_t M1 = std::max<int64_t>(M, 1);
        i0u.reserve(sizeof(int) * 3 * M1);
        fu.reserve(sizeof(T) * 3 * M1);
        i0s.reserve(sizeof(int) * 3 * M1);
        fs.reserve(sizeof(T) * 3 * M1);
        kw.reserve(sizeof(T) * 3 * M1 * ker.w);
        tile_of.reserve(sizeof(int) * M1);
        perm.reserve(sizeof(int) * M1);
        slot_id.reserve(sizeof(int) * M1);
        bin_start.reserve(sizeof(int) * (nb + 1));
        binmeta.reserve(sizeof(int) * (2 * (size_t)(nb + 1) + 1));  // counts | cursor | oob
        int *counts_p = binmeta.as<int>(), *cursor_p = counts_p + (nb + 1), *oob_p = cursor_p + (nb + 1);
        if (err_oob) oob_p = err_oob;
        oob_ptr = oob_p;
        FV_HIP(hipMemsetAsync(binmeta.p, 0, sizeof(int) * (2 * (size_t)(nb + 1) + 1), stream));
        BinArgs a{};
        a.w = ker.w;
        a.dim = dim;
        a.r2max = row_ext_ptr ? disc_radius * disc_radius * (1.0 + 2.0 * disc_margin()) : 0.0;  // only where the disc is used (build_block_order)
        for (int d = 0; d < 3; ++d) {
            a.xc[d] = geo.d[d].xc;
            a.invh[d] = 1.0 / geo.d[d].h;
            a.na[d] = d < dim ? geo.d[d].na : 1;
            a.nbin[d] = geo.nbin[d];
        }
        if (M > 0) {
            hipLaunchKernelGGL(k_bin_count<T>, dim3(cdiv(M, 256)), dim3(256), 0, stream, M, Mp, x,
                               y, z, a, i0u.as<int>(), fu.as<T>(), tile_of.as<int>(), counts_p,
                               oob_p);
        }
        exclusive_scan(counts_p, bin_start.as<int>(), nb);
        if (M > 0) {
            hipLaunchKernelGGL(k_bin_scatter_ids, dim3(cdiv(M, 256)), dim3(256), 0, stream, M, Mp,
                               (const int *)tile_of.as<int>(), (const int *)bin_start.as<int>(), cursor_p,
                               slot_id.as<int>());
            hipLaunchKernelGGL(k_bin_fill<T>, dim3(cdiv(M, 256)), dim3(256), 0, stream, M, Mp,
                               dim, i0u.as<int>(), fu.as<T>(), tile_of.as<int>(),
                               bin_start.as<int>(), (const int *)slot_id.as<int>(), i0s.as<int>(), fs.as<T>(),
                               perm.as<int>(), kw.as<T>(), ker.w, (T)ker.beta, (T)ker.c);
        }
    }

    // Exclusive scan of n device ints (n + 1 outputs) on the plan's stream.
    void exclusive_scan(const int *counts_p, int *out, int n) {
        if (n <= 4096) {
            hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, stream, counts_p, out, n);
            return;
        }
        const int nblk = (int)cdiv(n, 1024);
        FV_REQUIRE(nblk <= 1024 * 1024, "scan too large");
        scan_tot.reserve(sizeof(int) * (nblk + 1));
        scan_off.reserve(sizeof(int) * (nblk + 1));
        hipLaunchKernelGGL(k_scan_blocks, dim3(nblk), dim3(1024), 0, stream, counts_p, out,
                           scan_tot.as<int>(), n);
        if (nblk <= 4096) {
            hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, stream, scan_tot.as<int>(),
                               scan_off.as<int>(), nblk);
        } else {  // one more level (n > 4M)
            const int nb2 = (int)cdiv(nblk, 1024);
            scan_tot2.reserve(sizeof(int) * (nb2 + 1));
            scan_off2.reserve(sizeof(int) * (nb2 + 1));
            hipLaunchKernelGGL(k_scan_blocks, dim3(nb2), dim3(1024), 0, stream, scan_tot.as<int>(),
                               scan_off.as<int>(), scan_tot2.as<int>(), nblk);
            hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, stream, scan_tot2.as<int>(),
                               scan_off2.as<int>(), nb2);
            hipLaunchKernelGGL(k_scan_add, dim3(nb2), dim3(1024), 0, stream, scan_off.as<int>(),
                               scan_off2.as<int>(), nblk);
        }
        hipLaunchKernelGGL(k_scan_add, dim3(nblk), dim3(1024), 0, stream, out, scan_off.as<int>(), n);
    }

    // Use this plan only as a pruned 2-D FFT engine (type-1 path): geometry given directly.
    void set_fft_geometry(const DimGeom &gx, const DimGeom &gy) {
        DimGeom g[2] = {gx, gy};
        g[0].rm = true;
        g[1].rm = false;
        for (int d = 0; d < 2; ++d) {
            const bool same = geo.d[d].n2 == g[d].n2 && tw[d].p;
            geo.d[d] = g[d];
            if (same) continue;
            tw[d].reserve(sizeof(cplx<T>) * g[d].n2);
            hipLaunchKernelGGL(k_twiddle_table<T>, dim3(cdiv(g[d].n2, 256)), dim3(256), 0, stream,
                               g[d].n2, tw[d].as<cplx<T>>());
        }
        for (int d = 0; d < 2; ++d) tw_cur[d] = tw[d].as<cplx<T>>();
    }
    cplx<T> *fft_input(int ntrans) {
        int64_t c0, c1;
        buffer_cells(c0, c1);
        buf0.reserve(sizeof(cplx<T>) * c0 * ntrans);
        return buf0.as<cplx<T>>();
    }
    const cplx<T> *fft_output() const { return grid_out; }

    int out_of_box_count() {
        int v = 0;
        if (!oob_ptr) return 0;
        FV_HIP(hipMemcpyAsync(&v, oob_ptr, sizeof(int), hipMemcpyDeviceToHost, stream));
        FV_HIP(hipStreamSynchronize(stream));
        return v;
    }

    void strengths_buffer_reserve(int64_t cap, int ntrans) {
        strengths.reserve(sizeof(cplx<T>) * std::max<int64_t>(cap, 1) * ntrans);
    }
    cplx<T> *strengths_buffer(int ntrans) {
        strengths.reserve(sizeof(cplx<T>) * std::max<int64_t>(M, 1) * ntrans);
        return strengths.as<cplx<T>>();
    }

    // cin: device (ntrans, M) row-major in the caller's source order.
    void load_strengths(const cplx<T> *cin, int ntrans, int tpol, const double *scale_dev) {
        cplx<T> *cs = strengths_buffer(ntrans);
        if (M == 0) return;
        hipLaunchKernelGGL(k_load_strengths<T>, dim3(cdiv(M, 256)), dim3(256), 0, stream, M, Mp, ntrans,
                           tpol, dim, cin, perm.as<int>(), i0s.as<int>(), fs.as<T>(), geo.d[0].h,
                           geo.d[1].h, geo.d[2].h, geo.d[0].na, geo.d[1].na,
                           dim > 2 ? geo.d[2].na : 1, geo.d[0].btc, geo.d[1].btc, geo.d[2].btc,
                           scale_dev, cs);
    }

    // mate: a second plan with the same geometry, source capacity and transform count (another time
    // step of the same array); spread / fft then run both in one launch each (grid.y = 2)
    void spread(int ntrans, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, Nufft3 *mate = nullptr);
    template <int TCH>
    int launch_spread(int ntrans, int tbegin, hipEvent_t e0, hipEvent_t e1, Nufft3 *mate);
    void buffer_cells(int64_t &c0, int64_t &c1) const;
    void fft(int ntrans, Nufft3 *mate = nullptr);
    bool gang_compatible(const Nufft3 &o) const {
        bool ok = dim == 2 && o.dim == 2 && M == o.M && ker.w == o.ker.w;
        for (int d = 0; d < 2 && ok; ++d)
            ok = geo.d[d].na == o.geo.d[d].na && geo.d[d].n2 == o.geo.d[d].n2 && geo.d[d].no == o.geo.d[d].no &&
                 geo.d[d].h == o.geo.d[d].h && geo.d[d].xc == o.geo.d[d].xc && geo.d[d].btc == o.geo.d[d].btc;
        return ok;
    }
    double fft_traffic_cells() const;  // cells read + written by all FFT passes, per transform
    // real flops of the passes per transform, priced as plain radix-2 transforms: 5 n2 log2(n2) per line transformed
    // (x-pass: na_y lines, y-pass: the columns kept; 3-D adds the z lines) -- what the pruned passes replace
    double fft_flops() const {
        const DimGeom &x = geo.d[0], &y = geo.d[1], &z = geo.d[2];
        const double zin = dim > 2 ? z.na : 1;
        auto f = [](const DimGeom &g) { return 5.0 * g.n2 * std::log2((double)g.n2); };
        double c = zin * y.na * f(x) + zin * (double)xcols() * f(y);
        if (dim > 2 && !zdirect) c += (double)x.nos() * y.nos() * f(z);
        return c;
    }
    int64_t spread_cells() const {  // cells of A the spread writes, per transform
        return dim == 2 ? order_cells : row_ext_ptr ? order_cells * geo.d[2].na : geo.cells_a();
    }
    // Targets: base coordinates bt* (device, indexed by global baseline id), optional subset
    // index list / flip flags of length N, per-group scale (device, nfg doubles).
    void interp(int64_t N, const T *btx, const T *bty, const T *btz, const int *bl_idx,
                const signed char *flip, const double *scale_dev, int nfg, int tpol,
                cplx<T> *out, int64_t out_fg_stride, int64_t out_k_stride,
                const int64_t *out_pol_off, bool accumulate, const struct BasisTerm *basis = nullptr,
                int herm = 0, const int *ustart = nullptr, int64_t nuniq = 0, const int *upairs = nullptr,
                const struct WTerm *wt = nullptr);

   private:
    void rowfft(const cplx<T> *in, cplx<T> *out, const DimGeom &g, const cplx<T> *twd,
                int64_t nplanes, int64_t rpp, int64_t in_plane, int64_t in_row, int64_t in_elem,
                int64_t out_pitch = 0, int64_t rpp_valid = 0, const FusedArgs *fused = nullptr,
                const cplx<T> *in1 = nullptr, cplx<T> *out1 = nullptr, int in_blk = 0, int out_blk = 0);
    int b_block_log() const;  // column-blocked layout of the x-pass output (0: plain)
    bool y_reads_columns() const;  // the y-pass runs in column mode (else: tile transpose + row pass)
    int64_t b_pitch() const;  // row pitch of the x-pass output
    cplx<T> *grid_out = nullptr;  // where the last fft() left Ct
};

template <typename T>
template <int TCH>
int Nufft3<T>::launch_spread(int ntrans, int tbegin, hipEvent_t e0, hipEvent_t e1, Nufft3 *mate) {
    const int nchunk = (ntrans - tbegin) / TCH;
    if (nchunk == 0) return tbegin;
    const DimGeom &x = geo.d[0], &y = geo.d[1], &z = geo.d[2];
    // events (profiling only) ride on the dispatch itself: start on the first spread launch of a
    // transform batch, stop on the last -- no separate marker packets in the queue
    hipEvent_t es = tbegin == 0 ? e0 : nullptr;
    hipEvent_t ee = tbegin + nchunk * TCH == ntrans ? e1 : nullptr;
    if (dim == 2) {
        dim3 g((unsigned)((int64_t)order_n * nchunk), mate ? 2 : 1);  // the launch list: every 4-block group (inside the disc)
        SpreadMate sm{};
        if (mate) {
            sm.i0s = mate->i0s.template as<int>();
            sm.bin_start = mate->bin_start.template as<int>();
            sm.kw = mate->kw.p;
            sm.cs = mate->strengths.p;
            sm.grid = mate->buf0.p;
        }
        // lane mapping: channel groups once a block sees a few sources (M counts the catalog before
        // the horizon cut), lane per cell on sparse grids; FFTVIS_HIP_SPREAD_CELL = 1 / 0 forces one
        const char *force_cell = std::getenv("FFTVIS_HIP_SPREAD_CELL");
        const bool dense = (double)M >= 3.0 * (double)geo.nbin[0] * (double)geo.nbin[1];
        auto kern = k_spread2d<T, TCH>;
        if constexpr (TCH >= 8) {
            if (force_cell ? std::atoi(force_cell) == 0 : dense) kern = k_spread2d_cg<T, TCH>;
            // fp64 on source-dense grids: the accumulation as a matrix product (FFTVIS_HIP_SPREAD_MM = 0: channel groups)
            if constexpr (sizeof(T) == 8) {
                const char *emm = std::getenv("FFTVIS_HIP_SPREAD_MM");
                if (emm ? std::atoi(emm) != 0 : (dense && !force_cell)) kern = k_spread2d_mm<TCH>;
            }
        }
        hipExtLaunchKernelGGL(kern, g, dim3(SPREAD_THREADS), 0, stream, es, ee, 0, M,
                              (const int *)i0s.as<int>(), (const T *)kw.as<T>(),
                              (const int *)bin_start.as<int>(),
                              (const cplx<T> *)strengths.as<cplx<T>>(), ntrans, tbegin,
                              dec_cur[0], dec_cur[1],
                              buf0.as<cplx<T>>(), x.na, y.na, geo.nbin[0], ker.w,
                              order_ptr, nchunk, sm);
    } else {
        dim3 g((unsigned)cdiv(geo.nbin[0], 4), (unsigned)geo.nbin[1], (unsigned)(z.na * nchunk));
        hipExtLaunchKernelGGL((k_spread3d<T, TCH>), g, dim3(SPREAD_THREADS), 0, stream, es, ee, 0, M,
                              (const int *)i0s.as<int>(), (const T *)kw.as<T>(),
                              (const int *)bin_start.as<int>(),
                              (const cplx<T> *)strengths.as<cplx<T>>(), ntrans, tbegin, nchunk,
                              dec_cur[0], dec_cur[1],
                              dec_cur[2], buf0.as<cplx<T>>(), x.na, y.na, z.na,
                              geo.nbin[0], geo.nbin[1], ker.w, row_ext_ptr);
    }
    return tbegin + nchunk * TCH;
}

// Largest buffer (cells per transform) each ping-pong buffer has to hold during spread + fft.
// Row-FFT launch geometry for one dimension (shared by the launcher and the transpose decision):
// Q/8 threads per row (16..512), 256..512 threads per workgroup.
inline bool rowfft_uses_st(const DimGeom &g, bool col) {  // register-resident kernel applies
    // (a Q = 4096 column pass -- 2 columns per 512-thread workgroup, 120 VGPRs, two workgroups per CU -- was built
    // and is correct, but at 2.21 ms per launch it loses to tile transpose + row pass, 0.90 + 1.03 ms: 32-B column
    // segments cost four times the L1 / TA traffic of rows)
    static const int colmax = std::getenv("FFTVIS_HIP_COL_LOGQ_MAX") ? std::min(12, std::atoi(std::getenv("FFTVIS_HIP_COL_LOGQ_MAX"))) : 11;
    return g.logQ >= 9 && g.logQ <= (col ? colmax : 12) && !debug_switch_old_fft();
}
inline void rowfft_shape(const DimGeom &g, bool col, int &tpr, int &rpw) {
    if (rowfft_uses_st(g, col)) {  // Q/16 threads per row (64 for 512); 8 columns / 1-4 rows per workgroup
        tpr = g.logQ == 9 ? 64 : g.Q / 16;
        rpw = st_threads(g.logQ, col) / tpr;
        return;
    }
    tpr = 16;
    while (tpr < 512 && tpr < g.Q / FV_FFT_TPR_DIV1) tpr *= 2;
    rpw = std::max(1, FFT_THREADS / tpr);
}

// The y-pass reads 8 adjacent columns of B per workgroup when it can (column mode): B's rows are
// then padded to a multiple of 8 elements so that those 128-B segments are whole cache lines.
template <typename T>
int64_t Nufft3<T>::b_pitch() const {
    // whole 128-B lines per workgroup (8 columns) or per pair of neighbouring workgroups (4 columns each;
    // giving such pairs consecutive slots on one XCD was measured to change nothing: 1.836 vs 1.833 ms)
    return y_reads_columns() ? (xcols() + 7) / 8 * 8 : xcols();
}

// Column mode needs a kernel that holds >= 2 columns per workgroup and planes below 4 GiB per transform (its
// accesses are 32-bit byte offsets from a plane's base); larger planes take the tile transpose and a row pass.
template <typename T>
bool Nufft3<T>::y_reads_columns() const {
    const DimGeom &x = geo.d[0], &y = geo.d[1];
    int tpr, rpw;
    rowfft_shape(y, true, tpr, rpw);
    const int64_t pitch = (x.nos() + 7) / 8 * 8;
    return rpw >= 2 && std::max<int64_t>(y.na, y.nos()) * pitch * (int64_t)sizeof(cplx<T>) < (int64_t(1) << 32);
}

// B (x-pass output, y-pass input) in 64-byte column blocks when both passes run the register-resident kernels
// and the y-pass reads columns: the column pass then streams contiguous memory (DRAM pages, full lines) instead
// of one 64-B piece per row, and the x-pass writes 64-B pieces instead of whole runs.
template <typename T>
int Nufft3<T>::b_block_log() const {
    static const bool off = std::getenv("FFTVIS_HIP_NO_BLOCKED_B") != nullptr;
    const DimGeom &x = geo.d[0], &y = geo.d[1];
    int tpr, rpw;
    rowfft_shape(y, true, tpr, rpw);
    // (3-D: B sits between the x- and the y-pass of every (transform, z) plane exactly as in 2-D; FFTVIS_HIP_NO_BLOCKED_B3=1
    // keeps the plain planes there)
    static const bool off3 = std::getenv("FFTVIS_HIP_NO_BLOCKED_B3") != nullptr;
    if (off || (dim != 2 && off3) || !rowfft_uses_st(x, false) || !rowfft_uses_st(y, true) || !y_reads_columns()) return 0;
    static const int force = std::getenv("FFTVIS_HIP_B_BLOCK_LOG") ? std::atoi(std::getenv("FFTVIS_HIP_B_BLOCK_LOG")) : 0;
    if (force) return force;
    return sizeof(cplx<T>) == 16 && rpw < 8 ? 2 : 3;  // 64-B pieces (fp32, and 8-column workgroups: 8 elements)
}

template <typename T>
void Nufft3<T>::buffer_cells(int64_t &c0, int64_t &c1) const {
    const DimGeom &x = geo.d[0], &y = geo.d[1], &z = geo.d[2];
    const int64_t zin = dim > 2 ? z.na : 1, zout = dim > 2 ? (zdirect ? z.na : z.nos()) : 1;
    const int64_t A = zin * y.na * x.na, B = zin * y.na * b_pitch(), C = zin * x.nos() * y.nos(),
                  D = zout * x.nos() * y.nos();
    c0 = std::max({A, B, C, D});  // either buffer may end up holding any stage (transpose or not)
    c1 = c0;
}

template <typename T>
void Nufft3<T>::spread(int ntrans, hipEvent_t e0, hipEvent_t e1, Nufft3 *mate) {
    int64_t c0, c1;
    buffer_cells(c0, c1);
    buf0.reserve(sizeof(cplx<T>) * c0 * ntrans);
    if (mate) {
        FV_REQUIRE(gang_compatible(*mate), "gang launch needs two plans of one geometry");
        mate->buf0.reserve(sizeof(cplx<T>) * c0 * ntrans);
    }
    // whole chunks of 16 transforms per thread, then the binary remainder (<= 4 more launches)
    int t = launch_spread<16>(ntrans, 0, e0, e1, mate);
    t = launch_spread<8>(ntrans, t, e0, e1, mate);
    t = launch_spread<4>(ntrans, t, e0, e1, mate);
    t = launch_spread<2>(ntrans, t, e0, e1, mate);
    launch_spread<1>(ntrans, t, e0, e1, mate);
}

// rows = nplanes * rpp; element ia of row (plane, k) sits at plane*in_plane + k*in_row + ia*in_elem.
template <typename T>
void Nufft3<T>::rowfft(const cplx<T> *in, cplx<T> *out, const DimGeom &g, const cplx<T> *twd,
                       int64_t nplanes, int64_t rpp, int64_t in_plane, int64_t in_row,
                       int64_t in_elem, int64_t out_pitch, int64_t rpp_valid, const FusedArgs *fused,
                       const cplx<T> *in1, cplx<T> *out1, int in_blk, int out_blk) {
    static const int plans[9][4] = {{4, 0, 0, 0}, {3, 2, 0, 0}, {3, 3, 0, 0}, {4, 3, 0, 0}, {4, 4, 0, 0},
                                    {3, 3, 3, 0}, {4, 3, 3, 0}, {4, 4, 3, 0}, {4, 4, 4, 0}};  // logQ = 4 .. 12
    FV_REQUIRE(g.logQ >= 4 && g.logQ <= FFT_QMAX_LOG, "row FFT length out of range");
    RowDifArgs a{};
    a.n_in = g.na;
    a.n_out = g.no;
    a.n2 = g.n2;
    a.P = g.P;
    a.Q = g.Q;
    a.logQ = g.logQ;
    a.npass = 0;
    for (int s = 0; s < 4; ++s) {
        a.radix_log[s] = plans[g.logQ - 4][s];
        if (a.radix_log[s]) ++a.npass;
    }
    a.lds_row = fft_pidx(g.Q) | 1;
    a.colmode = in_elem != 1;
    rowfft_shape(g, a.colmode != 0, a.tpr, a.rpw);
    a.nrows = nplanes * rpp;
    a.rpp = rpp;
    a.rpp_valid = rpp_valid ? rpp_valid : rpp;
    a.in_plane = in_plane;
    a.in_row = in_row;
    a.in_elem = in_elem;
    a.out_pitch = out_pitch ? out_pitch : g.nos();
    a.cnt = g.sP() > 1 ? g.cnt() : 0;
    a.in_blk = in_blk;
    a.out_blk = out_blk;
    const int planes_per_trans = dim > 2 ? geo.d[2].na : 1;  // (3-D, direct third dimension: the z-planes of a transform share its plan)
    if (a.colmode && in_blk && col_tab && col_omask && !fused && g.logQ <= 11) {  // the y-pass of a column plan
        a.omask = col_omask;
        a.omask_tpol = col_tab_tpol * planes_per_trans;
        a.omask_nblk = col_omask_nblk;
    }
    if (first_pass_ext) a.row_ext = first_pass_ext;  // (row mode, register-resident kernels: fft() only sets it there)
    if (out_blk && col_tab) {  // the x-pass of a column plan
        a.ctab = col_xtab;
        a.ctab_stride = g.nos();
        a.ctab_tpol = col_tab_tpol * planes_per_trans;
    }
    FV_REQUIRE((!in_blk && !out_blk) || rowfft_uses_st(g, a.colmode != 0), "blocked planes: register-resident passes only");
    // column-mode and blocked accesses are 32-bit byte offsets from a plane's base (buffer descriptors)
    if (rowfft_uses_st(g, a.colmode != 0) && (a.colmode || in_blk || out_blk))
        FV_REQUIRE(std::max(in_plane, a.rpp_valid * a.out_pitch) * (int64_t)sizeof(cplx<T>) < (int64_t(1) << 32),
                   "grid planes of 4 GiB and more per transform are not supported by the column pass");
    // PAIR: two residues per job (k_rowfft_st) for folded rows of Q = 1024 / 2048 -- FFTVIS_HIP_PAIR = 0 off, 1 row mode
    // only, 2 row and column mode
    static const int pair_mode = std::getenv("FFTVIS_HIP_PAIR") ? std::atoi(std::getenv("FFTVIS_HIP_PAIR")) : FV_PAIR_DEFAULT;
    const bool st = rowfft_uses_st(g, a.colmode != 0);
    const bool pair = st && !fused && a.n_in > g.Q && g.P >= 2 && (g.logQ == 10 || g.logQ == 11) &&
                      (a.colmode ? pair_mode >= 2 : (pair_mode >= 1 && a.rpw >= 2));
    const int nl = pair && !a.colmode ? a.rpw / 2 : a.rpw;   // data lines per workgroup
    const int pj = pair ? (g.P + 1) / 2 : g.P;               // jobs per line group
    const int64_t ngroups8 = cdiv(cdiv(a.nrows, nl), 8);  // line groups, in eights (one per XCD)
    if (std::getenv("FFTVIS_HIP_DEBUG_FFT"))
        std::fprintf(stderr, "rowfft col=%d n_in=%d n_out=%d n2=%d P=%d Q=%d nrows=%lld rpw=%d pair=%d wgs=%lld\n", a.colmode, a.n_in,
                     a.n_out, a.n2, a.P, a.Q, (long long)a.nrows, a.rpw, (int)pair, (long long)(ngroups8 * 8 * pj));
    if (in1 && !st) {  // no gang variant of the LDS kernel: two launches
        rowfft(in, out, g, twd, nplanes, rpp, in_plane, in_row, in_elem, out_pitch, rpp_valid, fused);
        FusedArgs f1{};
        if (fused) {
            f1 = *fused;
            f1.out = fused->out1;
        }
        rowfft(in1, out1, g, twd, nplanes, rpp, in_plane, in_row, in_elem, out_pitch, rpp_valid, fused ? &f1 : nullptr);
        return;
    }
    a.in1 = in1;
    a.out1 = out1;
    dim3 jobs((unsigned)(ngroups8 * 8 * pj), in1 ? 2 : 1);
    a.jobs_per_xcd = 0;
    if (st && !a.colmode) {
        FV_REQUIRE(a.nrows < (int64_t(1) << 30), "row FFT: too many rows");
        a.jobs_per_xcd = (int)(cdiv(cdiv(a.nrows, 8), nl) * nl);           // rows per XCD, whole workgroups
        jobs.x = (unsigned)(8 * (a.jobs_per_xcd / nl) * pj);
    }
    if (rowfft_uses_st(g, a.colmode != 0)) {
        const int s1 = g.Q / (g.logQ == 9 ? 8 : 16);       // stride of the first radix pass
        // possibly non-zero inputs per thread: the row is centred (see the kernel), so the elements sit in
        // the first ceil(n_in / 2) and the last n_in / 2 slots -- NLD / 2 registers at either end
        const int need = a.n_in > g.Q ? 16 : 2 * (int)cdiv(a.n_in - a.n_in / 2, s1);
        const int nld = need <= 4 ? 4 : need <= 8 ? 8 : 16;
        const bool col = a.colmode != 0;
        FV_REQUIRE(!col || in_elem < (1 << 23), "column pass: row pitch beyond the 24-bit index multiply");
#define FV_ST_LAUNCH1(LQ, COLM, NLD, FUSEDV, FOLDV, INBV, PAIRV, FZ)                                     \
    hipLaunchKernelGGL((k_rowfft_st<T, LQ, COLM, NLD, FUSEDV, FOLDV, INBV, PAIRV>), jobs,              \
                       dim3(st_threads(LQ, COLM, PAIRV)), 0, stream, in, out, twd, a, FZ);
#define FV_ST_LAUNCH(LQ, COLM, NLD, FUSEDV, FOLDV, FZ)                                                  \
    {                                                                                                  \
        constexpr bool PAIRABLE = FOLDV && !FUSEDV && (LQ == 10 || LQ == 11);                          \
        if (COLM && in_blk && (1 << in_blk) == nl) {                                                   \
            if (PAIRABLE && pair) {                                                                    \
                FV_ST_LAUNCH1(LQ, COLM, NLD, FUSEDV, FOLDV, (COLM ? 2 : 0), PAIRABLE, FZ)              \
            } else {                                                                                   \
                FV_ST_LAUNCH1(LQ, COLM, NLD, FUSEDV, FOLDV, (COLM ? 2 : 0), false, FZ)                 \
            }                                                                                          \
        } else if (COLM && in_blk) {                                                                   \
            if (PAIRABLE && pair) {                                                                    \
                FV_ST_LAUNCH1(LQ, COLM, NLD, FUSEDV, FOLDV, (COLM ? 1 : 0), PAIRABLE, FZ)              \
            } else {                                                                                   \
                FV_ST_LAUNCH1(LQ, COLM, NLD, FUSEDV, FOLDV, (COLM ? 1 : 0), false, FZ)                 \
            }                                                                                          \
        } else {                                                                                       \
            if (PAIRABLE && pair) {                                                                    \
                FV_ST_LAUNCH1(LQ, COLM, NLD, FUSEDV, FOLDV, 0, PAIRABLE, FZ)                           \
            } else {                                                                                   \
                FV_ST_LAUNCH1(LQ, COLM, NLD, FUSEDV, FOLDV, 0, false, FZ)                              \
            }                                                                                          \
        }                                                                                              \
    }
#define FV_ST_GO(LQ, COLM, NLD)                                                                        \
    if (a.n_in > g.Q && NLD == (LQ == 9 ? 8 : 16)) {                                                   \
        FV_ST_LAUNCH(LQ, COLM, (LQ == 9 ? 8 : 16), false, true, FusedArgs{})                           \
    } else {                                                                                           \
        bool launched = false;                                                                         \
        if constexpr (COLM && LQ <= 10) { /* the fused gather rides on 8-column passes only */          \
            if (fused) {                                                                               \
                FV_ST_LAUNCH(LQ, COLM, NLD, COLM, false, *fused)                                       \
                launched = true;                                                                       \
            }                                                                                          \
        }                                                                                              \
        if (!launched) {                                                                               \
            FV_ST_LAUNCH(LQ, COLM, NLD, false, false, FusedArgs{})                                     \
        }                                                                                              \
    }
#define FV_ST_NLD(LQ, COLM)                                                                            \
    if (nld == 4) {                                                                                    \
        FV_ST_GO(LQ, COLM, 4)                                                                          \
    } else if (nld == 8 || LQ == 9) {                                                                  \
        FV_ST_GO(LQ, COLM, 8)                                                                          \
    } else {                                                                                           \
        FV_ST_GO(LQ, COLM, (LQ == 9 ? 8 : 16))                                                         \
    }
        if (g.logQ == 9) {
            if (col) { FV_ST_NLD(9, true) } else { FV_ST_NLD(9, false) }
        } else if (g.logQ == 10) {
            if (col) { FV_ST_NLD(10, true) } else { FV_ST_NLD(10, false) }
        } else if (g.logQ == 11) {
            if (col) { FV_ST_NLD(11, true) } else { FV_ST_NLD(11, false) }
        } else {
            if (col) { FV_ST_NLD(12, true) } else { FV_ST_NLD(12, false) }
        }
#undef FV_ST_NLD
#undef FV_ST_GO
#undef FV_ST_LAUNCH
#undef FV_ST_LAUNCH1
        return;
    }
    const size_t smem = sizeof(cplx<T>) * (size_t)a.lds_row * a.rpw;
    if (smem > 48 * 1024)  // per device, so not cached in a process-wide flag (handles may sit on several GPUs)
        FV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rowfft_dif<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_rowfft_dif<T>, jobs, dim3(a.tpr * a.rpw), smem, stream, in, out, twd, a);
}

// Algorithmic traffic of fft(): every pass reads its input once and writes its pruned output once
// (plus the explicit transpose when the y-pass cannot read columns directly).
template <typename T>
double Nufft3<T>::fft_traffic_cells() const {
    const DimGeom &x = geo.d[0], &y = geo.d[1], &z = geo.d[2];
    const double zin = dim > 2 ? z.na : 1;
    int tpr, rpw;
    rowfft_shape(y, true, tpr, rpw);
    const double xo = col_tab ? (double)col_ncc : (double)x.no;          // columns stored by the x-pass (column plan: those a target reads)
    const double ain = row_ext_ptr ? (double)order_cells : (double)x.na * y.na;  // cells of A inside the source disc
    double c = zin * (ain + xo * y.na);                                  // x-pass
    if (!y_reads_columns()) c += zin * 2.0 * x.no * y.na;                 // transpose
    c += zin * (xo * y.na + (last_fft_fused ? 0.0 : col_tab && col_omask && col_out_cells > 0 ? col_out_cells : xo * y.no));  // y-pass (no C when fused)
    if (dim > 2 && !zdirect) c += (double)x.no * y.no * (z.na + z.no);   // z-pass
    return c;
}

template <typename T>
void Nufft3<T>::fft(int ntrans, Nufft3 *mate) {
    const DimGeom &x = geo.d[0], &y = geo.d[1], &z = geo.d[2];
    int64_t c0, c1;
    buffer_cells(c0, c1);
    buf1.reserve(sizeof(cplx<T>) * c1 * ntrans);
    cplx<T> *cur = buf0.as<cplx<T>>(), *oth = buf1.as<cplx<T>>();
    cplx<T> *cur1 = nullptr, *oth1 = nullptr;
    if (mate) {
        FV_REQUIRE(gang_compatible(*mate), "gang launch needs two plans of one geometry");
        mate->buf1.reserve(sizeof(cplx<T>) * c1 * ntrans);
        cur1 = mate->buf0.template as<cplx<T>>();
        oth1 = mate->buf1.template as<cplx<T>>();
    }
    const int64_t zin = dim > 2 ? z.na : 1;       // planes per transform before the z-pass
    const int64_t np = (int64_t)ntrans * zin;     // (trans, z) planes
    // x-pass: A [p][na_y][na_x] -> B [p][na_y][xp]   (xp = no_x, padded to 8 for column mode)
    const int64_t xp = b_pitch();
    const int blk = b_block_log();
    FV_REQUIRE(!col_tab || (blk && y_reads_columns() && !fused_active), "column plan: blocked B, column-mode y-pass, stand-alone gather");
    first_pass_ext = row_ext_ptr;  // the spread left the blocks outside the disc unwritten (build_block_order)
    rowfft(cur, oth, x, tw_cur[0], np, y.na, (int64_t)y.na * x.na, x.na, 1, xp, 0, nullptr, cur1, oth1, 0, blk);
    first_pass_ext = nullptr;
    std::swap(cur, oth);
    std::swap(cur1, oth1);
    if (y_reads_columns()) {
        // the y-pass reads rpw adjacent columns of B at once (32-128 B segments; neighbouring workgroups share lines),
        // which fuses the transpose:  B -> C [p][no_x][no_y]; the xp - no_x padding columns of a
        // plane are skipped as rows
        rowfft(cur, oth, y, tw_cur[1], np, xp, (int64_t)y.na * xp, 1, xp, 0, xcols(),
               fused_active ? &fused_args : nullptr, cur1, oth1, blk, 0);
        std::swap(cur, oth);
        std::swap(cur1, oth1);
    } else {
        // long columns: explicit tile transpose B -> Bt [p][no_x][na_y], then contiguous rows
        dim3 tg((unsigned)cdiv(x.nos(), 32), (unsigned)cdiv(y.na, 32), (unsigned)np);
        hipLaunchKernelGGL(k_transpose<T>, tg, dim3(256), 0, stream, cur, oth, y.na, x.nos());
        if (mate) hipLaunchKernelGGL(k_transpose<T>, tg, dim3(256), 0, stream, cur1, oth1, y.na, x.nos());
        rowfft(oth, cur, y, tw_cur[1], np, x.nos(), (int64_t)x.nos() * y.na, y.na, 1, 0, 0, nullptr,
               oth1, cur1);
    }
    if (dim > 2 && !zdirect) {
        // z-pass: C [t][na_z][nc] (nc = no_x no_y) -> D [t][nc][no_z]; adjacent (lx, ly) columns
        // are adjacent in memory, so the column-mode load is coalesced whenever rpw >= 4.
        const int64_t nc = (int64_t)x.nos() * y.nos();
        rowfft(cur, oth, z, tw_cur[2], ntrans, nc, (int64_t)z.na * nc, 1, nc);
        std::swap(cur, oth);
    }
    grid_out = cur;
    last_fft_fused = fused_active;
    if (mate) {
        mate->grid_out = cur1;
        mate->last_fft_fused = fused_active;
    }
    fused_active = false;
}

template <typename T>
bool Nufft3<T>::fused_possible() const {
    static const bool off = std::getenv("FFTVIS_HIP_NO_FUSED_GATHER") != nullptr;
    if (off || dim != 2) return false;
    const DimGeom &x = geo.d[0], &y = geo.d[1];
    int tpr, rpw;
    rowfft_shape(y, true, tpr, rpw);
    if (!rowfft_uses_st(y, true) || rpw != 8 || y.logQ > 10) return false;   // column-mode last pass, 8 columns per workgroup
    if (x.sP() != 1 || y.P != 1) return false;                 // natural order in both dimensions
    const int row_slots = y.logQ == 9 ? 577 : 1153;            // StPlan<9|10, true>::ROW
    return 2 * y.no <= row_slots && b_pitch() % 8 == 0;        // the 8 x n_out tile fits the exchange buffers
}

template <typename T>
bool Nufft3<T>::prepare_fused_gather(int64_t N, const T *btx, const T *bty, const int *bl_idx,
                                     const signed char *flip, const double *scale_dev, int nfg, int tpol,
                                     cplx<T> *out, int64_t out_fg_stride, int64_t out_k_stride,
                                     const int64_t *out_pol_off, int64_t targets_serial, cplx<T> *out_mate) {
    if (N == 0 || nfg == 0 || tpol > 4 || !fused_possible()) return false;
    const DimGeom &x = geo.d[0], &y = geo.d[1];
    FusedKey key;
    key.serial = geom_serial;
    key.targets = targets_serial;
    key.N = N;
    key.ofs = out_fg_stride;
    key.oks = out_k_stride;
    key.btx = btx;
    key.bl_idx = bl_idx;
    key.flip = flip;
    key.scale = scale_dev;
    key.nfg = nfg;
    key.tpol = tpol;
    const double kg[10] = {x.h, y.h, x.btc, y.btc, x.xc, y.xc, (double)x.no, (double)y.no, (double)x.n2, (double)y.n2};
    for (int i = 0; i < 10; ++i) key.g[i] = kg[i];
    FgGeom gm{};
    gm.w = ker.w;
    gm.nox = x.no;
    gm.noy = y.no;
    gm.n2x = x.n2;
    gm.n2y = y.n2;
    gm.hx = x.h;
    gm.hy = y.h;
    gm.btcx = x.btc;
    gm.btcy = y.btc;
    gm.xcx = x.xc;
    gm.xcy = y.xc;
    gm.out_fg_stride = out_fg_stride;
    gm.out_k_stride = out_k_stride;
    gm.rec = (int)((sizeof(FgHdr) + 2 * (size_t)ker.w * sizeof(T) + 15) / 16 * 16);
    gm.ngx = (int)(b_pitch() / 8);
    const int nb = nfg * gm.ngx;
    if (!(key == fused_key)) {
        const int64_t items = N * nfg;
        fg_recs.reserve((size_t)gm.rec * items);
        fg_meta.reserve(sizeof(int) * 2 * (size_t)(nb + 1));
        fg_start.reserve(sizeof(int) * (size_t)(nb + 1));
        int *counts_p = fg_meta.as<int>(), *cursor_p = counts_p + (nb + 1);
        FV_HIP(hipMemsetAsync(fg_meta.p, 0, sizeof(int) * 2 * (size_t)(nb + 1), stream));
        const dim3 gb((unsigned)cdiv(items, 256));
        hipLaunchKernelGGL((k_fg_build<T, 0>), gb, dim3(256), 0, stream, N, nfg, btx, bty, bl_idx, flip, scale_dev,
                           gm, ker, fg_recs.as<unsigned char>(), counts_p, (const int *)nullptr, cursor_p,
                           (int *)nullptr);
        exclusive_scan(counts_p, fg_start.as<int>(), nb);
        // every item sits in at most (w + 14) / 8 column groups
        fg_list.reserve(sizeof(int) * (size_t)items * ((ker.w + 14) / 8));
        hipLaunchKernelGGL((k_fg_build<T, 1>), gb, dim3(256), 0, stream, N, nfg, btx, bty, bl_idx, flip, scale_dev,
                           gm, ker, fg_recs.as<unsigned char>(), counts_p, (const int *)fg_start.as<int>(),
                           cursor_p, fg_list.as<int>());
        fused_key = key;
    }
    fused_args.lstart = fg_start.as<int>();
    fused_args.list = fg_list.as<int>();
    fused_args.recs = fg_recs.as<unsigned char>();
    fused_args.rec = gm.rec;
    fused_args.ngx = gm.ngx;
    fused_args.tpol = tpol;
    fused_args.w = ker.w;
    fused_args.out = out;
    fused_args.out1 = out_mate;
    for (int r = 0; r < 4; ++r) fused_args.pol_off[r] = out_pol_off ? out_pol_off[r] : 0;
    fused_args.tflip = transpose_flipped ? 1 : 0;
    fused_active = true;
    return true;
}

template <typename T>
void Nufft3<T>::interp(int64_t N, const T *btx, const T *bty, const T *btz, const int *bl_idx,
                       const signed char *flip, const double *scale_dev, int nfg, int tpol,
                       cplx<T> *out, int64_t out_fg_stride, int64_t out_k_stride,
                       const int64_t *out_pol_off, bool accumulate, const BasisTerm *basis, int herm,
                       const int *ustart, int64_t nuniq, const int *upairs, const WTerm *wt) {
    if (N == 0 || nfg == 0) return;
    FV_REQUIRE(!upairs || (herm && ustart), "paired runs: packed gathers over run lists");
    if (ustart) N = nuniq;  // items are the distinct targets; bl_idx / flip stay the caller's full list
    FV_REQUIRE(!herm || (tpol == 2 && (herm == 2 || !basis || basis->kk == basis->ll)),
               "packed gather: two transforms per frequency; off-diagonal eigenbeam terms only with real strengths");
    for (int d = 0; d < dim && herm; ++d) FV_REQUIRE(geo.d[d].btc == 0.0, "packed gather needs a box symmetric about 0");
    InterpArgs a{};
    const cplx<T> *coef = nullptr;
    const int *ant1 = nullptr, *ant2 = nullptr;
    if (basis) {
        a.basis = 1;
        a.kk = basis->kk;
        a.ll = basis->ll;
        a.nbasis = basis->nbasis;
        a.ncoef_freq = basis->nfreq;
        a.f_first = basis->f_first;
        a.basis_part = basis->part;
        a.negate_all = basis->negate;
        coef = (const cplx<T> *)basis->coef;
        ant1 = basis->ant1;
        ant2 = basis->ant2;
    }
    a.w = ker.w;
    a.tpol = tpol;
    a.nfg = nfg;
    for (int i = 0; i < 3; ++i) a.P[i] = a.cnt[i] = 1;
    // grid dimensions from fastest to slowest: 2-D (y, x), 3-D (z, y, x); direct third dimension: (y, x) per z-plane
    const int map2[3] = {1, 0, 0}, map3[3] = {2, 1, 0};
    const bool zd = dim == 3 && zdirect;
    const int gdim = zd ? 2 : dim;
    const int *map = gdim == 2 ? map2 : map3;
    const T *bts[3] = {btx, bty, btz};
    const T *bt[3] = {nullptr, nullptr, nullptr};
    if (zd) {
        FV_REQUIRE(btz, "3-D targets need their third coordinate");
        a.zd_n = geo.d[2].na;
        a.zd_h = geo.d[2].h;
        a.zd_btc = geo.d[2].btc;
        a.zd_xc = geo.d[2].xc;
        bt[2] = btz;
    }
    for (int i = 0; i < gdim; ++i) {
        const DimGeom &g = geo.d[map[i]];
        a.n2[i] = g.n2;
        a.no[i] = g.no;
        a.P[i] = g.sP();
        a.cnt[i] = g.cnt();
        a.h[i] = g.h;
        a.btc[i] = g.btc;
        a.xc[i] = g.xc;
        bt[i] = bts[map[i]];
    }
    // the gather addresses a footprint's rows with 32-bit element offsets inside one (x, y) slab of a transform
    FV_REQUIRE((int64_t)a.P[0] * a.cnt[0] * (col_tab ? (int64_t)col_ncc : (int64_t)a.P[1] * a.cnt[1]) < ((int64_t)1 << 31),
               "transform output slab of 2^31 elements or more: beyond the gather's 32-bit row offsets");
    a.out_fg_stride = out_fg_stride;
    a.out_k_stride = out_k_stride;
    for (int r = 0; r < 16; ++r) a.out_pol_off[r] = out_pol_off ? out_pol_off[r] : 0;
    a.accumulate = accumulate ? 1 : 0;
    if (col_tab) {
        FV_REQUIRE(gdim == 2, "column plan: 2-D transforms (or 3-D ones with the direct third dimension)");
        a.ctab = col_tab;
        a.ctab_stride = geo.d[0].nos();
        a.ncc = col_ncc;
        a.err = col_err;
    }
    a.herm = herm;
    a.transpose_flipped = transpose_flipped ? 1 : 0;
    a.wt_k = -1;
    if (wt) {
        FV_REQUIRE(dim == 2 && wt->bz && wt->k >= 0, "height terms ride on 2-D transforms");
        a.wt_k = wt->k;
        a.wt_zc = wt->zc;
        a.wt_zh = wt->zh;
        a.wt_bz = wt->bz;
    }
    const int64_t items = N * nfg;
    constexpr int IPW = INTERP_THREADS / GROUP;
    a.items_per_xcd = cdiv(cdiv(items, 8), IPW) * IPW;  // whole workgroups
    const dim3 grid((unsigned)(8 * (a.items_per_xcd / IPW)));
    const bool r9 = ker.w <= 9;
    FV_REQUIRE(!(zd && wt), "height terms ride on 2-D runs, the direct third dimension on 3-D ones");
    auto kern = gdim == 2 ? (herm ? (r9 ? k_interp<T, 2, true, 9> : k_interp<T, 2, true, 16>)
                                 : (r9 ? k_interp<T, 2, false, 9> : k_interp<T, 2, false, 16>))
                         : (herm ? (r9 ? k_interp<T, 3, true, 9> : k_interp<T, 3, true, 16>)
                                 : (r9 ? k_interp<T, 3, false, 9> : k_interp<T, 3, false, 16>));
    if (zd)
        kern = herm ? (r9 ? k_interp<T, 2, true, 9, true, false> : k_interp<T, 2, true, 16, true, false>)
                    : (r9 ? k_interp<T, 2, false, 9, true, false> : k_interp<T, 2, false, 16, true, false>);
    if (wt)
        kern = herm ? (r9 ? k_interp<T, 2, true, 9, false, true> : k_interp<T, 2, true, 16, false, true>)
                    : (r9 ? k_interp<T, 2, false, 9, false, true> : k_interp<T, 2, false, 16, false, true>);
    hipLaunchKernelGGL(kern, grid, dim3(INTERP_THREADS), 0, stream, (const cplx<T> *)grid_out, N, bt[0], bt[1], bt[2],
                       bl_idx, flip, scale_dev, a, ker, out, coef, ant1, ant2, ustart, upairs);
}

}  // namespace fv
