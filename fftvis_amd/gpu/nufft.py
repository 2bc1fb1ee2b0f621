"""GPU type-3 NUFFTs: the filled-in twins of the reference stubs ``gpu_nufft2d`` /
``gpu_nufft3d`` (src/fftvis/gpu/nufft.py:11-50,53-98), with the semantics of
``cpu_nufft2d`` / ``cpu_nufft3d`` (src/fftvis/cpu/nufft.py:11-59,62-118):

    out[t, k] = sum_j weights[t, j] * exp(+i (u_k x_j + v_k y_j [+ w_k z_j]))

computed on the MI355X by libfftvis_hip (spread -> pruned row FFTs -> gather) to relative accuracy
``eps``.  Input dtype selects the precision exactly as finufft does (float32 -> complex64).
"""

from __future__ import annotations

import numpy as np

from .. import _lib


def _run(coords, weights, targets, eps, upsample_factor, direct=False, device=0):
    x0 = np.asarray(coords[0])
    prec = 1 if x0.dtype == np.float32 else 2
    rdt = np.float32 if prec == 1 else np.float64
    cdt = np.complex64 if prec == 1 else np.complex128
    X = [np.ascontiguousarray(a, dtype=rdt) for a in coords]
    S = [np.ascontiguousarray(a, dtype=rdt) for a in targets]
    w = np.asarray(weights)
    squeeze = w.ndim == 1
    c = np.ascontiguousarray(np.atleast_2d(w), dtype=cdt)
    d = len(X)
    M, N = X[0].size, S[0].size
    if any(a.shape != (M,) for a in X) or c.shape[1] != M:
        raise ValueError("source coordinates and weights must share their last axis")
    if any(a.shape != (N,) for a in S):
        raise ValueError("target coordinate arrays must have equal length")
    out = np.empty((c.shape[0], N), dtype=cdt)
    X += [None] * (3 - d)
    S += [None] * (3 - d)
    L = _lib.lib()
    _lib.require_gpu()
    if direct:
        st = L.fv_nudft3_direct(device, prec, d, M, _lib.ptr(X[0]), _lib.ptr(X[1]), _lib.ptr(X[2]),
                                _lib.ptr(c), c.shape[0], N, _lib.ptr(S[0]), _lib.ptr(S[1]),
                                _lib.ptr(S[2]), _lib.ptr(out))
    else:
        st = L.fv_nufft3(device, prec, d, M, _lib.ptr(X[0]), _lib.ptr(X[1]), _lib.ptr(X[2]),
                         _lib.ptr(c), c.shape[0], N, _lib.ptr(S[0]), _lib.ptr(S[1]),
                         _lib.ptr(S[2]), float(eps), float(upsample_factor), _lib.ptr(out))
    _lib.check(st)
    return out[0] if squeeze else out


def gpu_nufft2d(x, y, weights, u, v, eps, n_threads: int = 1, upsample_factor=2):
    """2-D type-3 NUFFT on the GPU (n_threads is accepted and ignored, gpu/nufft.py:38)."""
    return _run([x, y], weights, [u, v], eps, upsample_factor)


def gpu_nufft3d(x, y, z, weights, u, v, w, eps, n_threads: int = 1, upsample_factor=2):
    """3-D type-3 NUFFT on the GPU."""
    return _run([x, y, z], weights, [u, v, w], eps, upsample_factor)


def gpu_nudft_direct(coords, weights, targets, device=0):
    """Brute-force O(MN) sum on the GPU -- an independent checker, not a product path."""
    return _run(list(coords), weights, list(targets), 0.0, 2, direct=True, device=device)
