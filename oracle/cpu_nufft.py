"""CPU type-3 NUFFT (numpy + scipy.fft) -- TEST INFRASTRUCTURE / CPU BASELINE.

A CPU port of the *published* FINUFFT type-3 algorithm (Barnett, Magland &
af Klinteberg, SISC 2019: ES kernel, spread -> deconvolve -> FFT -> interp ->
deconvolve), i.e. the algorithm the reference invokes through
``finufft.nufft2d3/nufft3d3`` at src/fftvis/cpu/nufft.py:48-59,105-118 with
``upsampfac=2``.  finufft itself is not available in this pipeline, so this is
labelled "port" wherever it is timed (bench.py cpu_baseline.kind) and is NOT
the parity oracle -- that is the exact sum in fftvis_oracle.nudft_type3 /
nudft.c, against which this port is itself checked (tests/test_oracle_golden.py).

Only tests/ and bench.py's cpu_baseline leg may import this.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import scipy.fft as sfft

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def _clib():
    """oracle/libcpunufft.so (C/OpenMP spread + interp); built on demand with oracle/Makefile."""
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "libcpunufft.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libcpunufft.so"])
        _CLIB = ctypes.CDLL(path)
        _CLIB.cn_spread2d.restype = None
        _CLIB.cn_interp2d.restype = None
    return _CLIB


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _ncores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _nthreads(work: int) -> int:
    """OpenMP threads for a spread/interp call: every core this process may run on once each gets
    >= 2.5e4 kernel-cell updates (a team costs ~10 us to fork; below that the call is over sooner on
    fewer threads) -- all cores for every BASELINE configuration from C2 up."""
    return int(max(1, min(_ncores(), work // 25_000)))


def next235even(n: int) -> int:
    n = max(2, int(n))
    n += n % 2
    while True:
        m = n
        for p in (2, 3, 5):
            while m % p == 0:
                m //= p
        if m == 1:
            return n
        n += 2


def es_params(eps: float, sigma: float = 2.0):
    if sigma == 2.0:
        w = int(np.ceil(np.log10(10.0 / eps)))
    else:
        w = min(15, int(np.ceil(-np.log(eps) / (np.pi * np.sqrt(1 - 1 / sigma)))) + 1)  # as fv_eskernel.h
    w = max(2, min(16, w))
    bow = {2: 2.20, 3: 2.26, 4: 2.38}.get(w, 2.30)
    if sigma != 2.0:
        bow = 0.97 * np.pi * (1 - 1 / (2 * sigma))
    return w, bow * w


def es_eval(z, w, beta):
    t = 1.0 - (2.0 * z / w) ** 2
    out = np.zeros_like(z)
    ok = t > 0
    out[ok] = np.exp(beta * (np.sqrt(t[ok]) - 1.0))
    return out


def es_hat(theta, w, beta, nq=None):
    """psi_hat(theta) = int psi(xi) cos(theta xi) dxi by Gauss-Legendre."""
    nq = nq or (4 + 2 * w)
    z, om = np.polynomial.legendre.leggauss(2 * nq)
    z, om = z[nq:], om[nq:]
    f = w * om * np.exp(beta * (np.sqrt(1 - z * z) - 1))
    return np.cos(np.multiply.outer(theta, 0.5 * w * z)) @ f


def _geom(X, S, sigma, w):
    Xs, Ss = X, S
    if X == 0:
        if S == 0:
            Xs = Ss = 1.0
        else:
            Xs = max(Xs, 1.0 / S)
    else:
        Ss = max(Ss, 1.0 / X)
    n1 = int(np.ceil(2 * sigma * Ss * Xs / np.pi + w + 1))
    n1 += n1 % 2
    # the gather's footprints (|eta| <= n2 / (2 sigma) + w / 2) must stay inside the n2 outputs
    n2 = next235even(max(int(np.ceil(sigma * n1)), int(np.ceil((w + 4) / (1.0 - 1.0 / sigma)))))
    h = np.pi / (sigma * Ss)
    return n1, n2, h


def nufft_type3(coords, c, targets, eps=1e-9, sigma=2.0, isign=+1, workers=-1, use_c_kernels=True):
    """f[t,k] ~= sum_j c[t,j] exp(isign i s_k.x_j) to relative accuracy ~eps.

    coords / targets: lists of d (2 or 3) 1-D arrays; c: (M,) or (ntrans, M).
    """
    c = np.asarray(c)
    squeeze = c.ndim == 1
    c2 = np.atleast_2d(c).astype(complex)
    d = len(coords)
    X = [np.asarray(a, float) for a in coords]
    S = [np.asarray(a, float) * (1.0 if isign >= 0 else -1.0) for a in targets]
    w, beta = es_params(eps, sigma)
    xc = [0.5 * (a.min() + a.max()) for a in X]
    sc = [0.5 * (a.min() + a.max()) for a in S]
    Xh = [max(abs(a - m).max(), 0.0) for a, m in zip(X, xc)]
    Sh = [max(abs(a - m).max(), 0.0) for a, m in zip(S, sc)]
    n1, n2, h = zip(*[_geom(Xh[i], Sh[i], sigma, w) for i in range(d)])
    ntr, M = c2.shape
    N = S[0].size

    # pre-phase  c_j <- c_j exp(i s_c . x'_j)
    ph = sum(sc[i] * (X[i] - xc[i]) for i in range(d))
    cp = c2 * np.exp(1j * ph)

    # spread onto the centred n2 grid (index = m + n2/2), kernel psi_1
    i0, ker = [], []
    for i in range(d):
        p = (X[i] - xc[i]) / h[i] + n2[i] // 2
        a = np.ceil(p - w / 2).astype(np.int64)
        z = a[:, None] + np.arange(w)[None, :] - p[:, None]
        i0.append(a)
        ker.append(es_eval(z, w, beta))
    ar = np.arange(w)
    use_c = d == 2 and use_c_kernels
    if use_c:
        pxy = [np.ascontiguousarray((X[i] - xc[i]) / h[i] + n2[i] // 2) for i in range(2)]
        grid = np.empty((ntr, n2[1], n2[0]), dtype=complex)
        cpc = np.ascontiguousarray(cp)
        _clib().cn_spread2d(ctypes.c_int64(M), _vp(pxy[0]), _vp(pxy[1]), _vp(cpc), ctypes.c_int(ntr),
                            ctypes.c_int(w), ctypes.c_double(beta), ctypes.c_int(n2[0]),
                            ctypes.c_int(n2[1]), _vp(grid), ctypes.c_int(_nthreads(M * ntr * w * w)))
    else:
        grid = np.zeros((ntr,) + tuple(n2[::-1]), dtype=complex)  # [t][(z)][y][x]
    if use_c:
        pass
    elif d == 2:
        wt = ker[1][:, :, None] * ker[0][:, None, :]
        iy = (i0[1][:, None] + ar)[:, :, None] + np.zeros((1, 1, w), np.int64)
        ix = (i0[0][:, None] + ar)[:, None, :] + np.zeros((1, w, 1), np.int64)
        for t in range(ntr):
            np.add.at(grid[t], (iy, ix), cp[t][:, None, None] * wt)
    else:
        wt = ker[2][:, :, None, None] * ker[1][:, None, :, None] * ker[0][:, None, None, :]
        zz = np.zeros((1, w, w, w), np.int64)
        iz = (i0[2][:, None] + ar)[:, :, None, None] + zz
        iy = (i0[1][:, None] + ar)[:, None, :, None] + zz
        ix = (i0[0][:, None] + ar)[:, None, None, :] + zz
        for t in range(ntr):
            np.add.at(grid[t], (iz, iy, ix), cp[t][:, None, None, None] * wt)

    # deconvolve psi_2 on the grid (type-2 pre-correction) + centring signs
    for i in range(d):
        idx = np.arange(n2[i])
        m = idx - n2[i] // 2
        fac = np.zeros(n2[i])
        act = np.abs(m) <= n1[i] // 2
        fac[act] = 1.0 / es_hat(2 * np.pi * m[act] / n2[i], w, beta)
        fac *= 1.0 - 2.0 * (idx % 2)
        shape = [1] * (d + 1)
        shape[d - i] = n2[i]
        grid *= fac.reshape(shape)

    # unnormalised inverse-sign FFT, in place
    axes = tuple(range(1, d + 1))
    grid = sfft.ifftn(grid, axes=axes, norm="forward", workers=workers, overwrite_x=True)

    # interpolate at eta = theta n2 / (2 pi) + n2/2, theta = h s'
    j0, kq, dec = [], [], np.ones(N)
    for i in range(d):
        sp = S[i] - sc[i]
        eta = sp * h[i] * n2[i] / (2 * np.pi) + n2[i] // 2
        a = np.ceil(eta - w / 2).astype(np.int64)
        cols = a[:, None] + np.arange(w)[None, :]
        z = cols - eta[:, None]
        kq.append(es_eval(z, w, beta) * (1.0 - 2.0 * (cols % 2)))
        j0.append(a)
        dec *= es_hat(h[i] * sp, w, beta)
    sgn = 1.0
    for i in range(d):
        sgn *= 1.0 - 2.0 * ((n2[i] // 2) % 2)
    out = np.empty((ntr, N), dtype=complex)
    if use_c:
        eta = [np.ascontiguousarray((S[i] - sc[i]) * h[i] * n2[i] / (2 * np.pi) + n2[i] // 2) for i in range(2)]
        gridc = np.ascontiguousarray(grid)
        _clib().cn_interp2d(ctypes.c_int64(N), _vp(eta[0]), _vp(eta[1]), ctypes.c_int(ntr),
                            ctypes.c_int(w), ctypes.c_double(beta), ctypes.c_int(n2[0]),
                            ctypes.c_int(n2[1]), _vp(gridc), _vp(out), ctypes.c_int(_nthreads(N * ntr * w * w)))
    elif d == 2:
        iy = (j0[1][:, None] + ar)[:, :, None] + np.zeros((1, 1, w), np.int64)
        ix = (j0[0][:, None] + ar)[:, None, :] + np.zeros((1, w, 1), np.int64)
        wt = kq[1][:, :, None] * kq[0][:, None, :]
        for t in range(ntr):
            out[t] = (grid[t][iy, ix] * wt).sum(axis=(1, 2))
    else:
        zz = np.zeros((1, w, w, w), np.int64)
        iz = (j0[2][:, None] + ar)[:, :, None, None] + zz
        iy = (j0[1][:, None] + ar)[:, None, :, None] + zz
        ix = (j0[0][:, None] + ar)[:, None, None, :] + zz
        wt = kq[2][:, :, None, None] * kq[1][:, None, :, None] * kq[0][:, None, None, :]
        for t in range(ntr):
            out[t] = (grid[t][iz, iy, ix] * wt).sum(axis=(1, 2, 3))
    post = np.exp(1j * sum(S[i] * xc[i] for i in range(d))) * (sgn / dec)
    out *= post
    if isign < 0:
        pass  # targets were negated above: exp(-i s x) = exp(i (-s) x)
    return out[0] if squeeze else out
