"""GPU twin of the reference's ``gpu/utils.py`` (stub at src/fftvis/gpu/utils.py:8-22)."""

from __future__ import annotations

import numpy as np

from .. import _lib


def inplace_rot(rot: np.ndarray, b: np.ndarray, device: int = 0) -> None:
    """In place ``b[:, n] <- rot @ b[:, n]`` for ``b`` of shape (3, n), on the GPU
    (semantics of reference cpu/utils.py:5-24)."""
    if b.ndim != 2 or b.shape[0] != 3:
        raise ValueError("b must have shape (3, n)")
    prec = 1 if b.dtype == np.float32 else 2
    rdt = np.float32 if prec == 1 else np.float64
    work = np.ascontiguousarray(b, dtype=rdt)
    R = np.ascontiguousarray(rot, dtype=np.float64)
    _lib.require_gpu()
    _lib.check(_lib.lib().fv_inplace_rot(device, prec, _lib.ptr(R), _lib.ptr(work), work.shape[1]))
    b[...] = work
