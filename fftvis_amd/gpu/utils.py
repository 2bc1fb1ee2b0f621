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


def astrom_topo(eq: np.ndarray, astrom: np.ndarray, device: int = 0) -> np.ndarray:
    """(3, n) ICRS unit vectors -> (3, n) topocentric (east, north, up) unit vectors under ONE 31-double ERFA
    ``eraASTROM`` context, on the GPU (``fv_astrom_topo``): a time step of the device-side coordinate manager,
    what matvis' ``CoordinateRotationERFA.rotate(t)`` leaves in ``all_coords_topo`` (reference
    cpu_simulate.py:937)."""
    eq = np.asarray(eq)
    if eq.ndim != 2 or eq.shape[0] != 3:
        raise ValueError("eq must have shape (3, n)")
    prec = 1 if eq.dtype == np.float32 else 2
    work = np.ascontiguousarray(eq, dtype=np.float32 if prec == 1 else np.float64)
    ctx = np.ascontiguousarray(astrom, dtype=np.float64)
    if ctx.shape != (31,):
        raise ValueError("astrom must be one context of 31 float64")
    out = np.empty_like(work)
    _lib.require_gpu()
    _lib.check(_lib.lib().fv_astrom_topo(device, prec, _lib.ptr(ctx), work.shape[1], _lib.ptr(work), _lib.ptr(out)))
    return out
