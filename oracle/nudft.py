"""ctypes loader for oracle/libnudft.so -- TEST INFRASTRUCTURE (CPU oracle).

Exact fp64 direct type-3 NUDFT in C/OpenMP; same definition as
``fftvis_oracle.nudft_type3`` (which it is checked against in
tests/test_oracle_golden.py).  Stands in for finufft at
src/fftvis/cpu/nufft.py:48-59,105-118.  PARITY UNPINNED against finufft.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libnudft.so"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libnudft.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.nudft_type3_f64.restype = None
        _LIB.nudft_num_threads.restype = ctypes.c_int
    return _LIB


def num_threads():
    return int(_lib().nudft_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def nudft_type3(coords, c, targets, isign=+1):
    """f[t,k] = sum_j c[t,j] exp(isign*i*s_k.x_j); coords/targets are lists of d arrays."""
    c = np.asarray(c)
    squeeze = c.ndim == 1
    c2 = np.ascontiguousarray(np.atleast_2d(c), dtype=np.complex128)
    X = [np.ascontiguousarray(a, dtype=np.float64) for a in coords]
    S = [np.ascontiguousarray(a, dtype=np.float64) for a in targets]
    d = len(X)
    M, N = X[0].size, S[0].size
    out = np.empty((c2.shape[0], N), dtype=np.complex128)
    X += [None] * (3 - d)
    S += [None] * (3 - d)
    _lib().nudft_type3_f64(
        ctypes.c_int(d), ctypes.c_int64(M), _p(X[0]), _p(X[1]), _p(X[2]), _p(c2),
        ctypes.c_int(c2.shape[0]), ctypes.c_int64(N), _p(S[0]), _p(S[1]), _p(S[2]),
        ctypes.c_int(isign), _p(out),
    )
    return out[0] if squeeze else out
