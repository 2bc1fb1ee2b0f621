"""Lattice detection for antenna layouts: decides whether the baselines of a flat array are
integer combinations of two basis vectors, which lets the engine use the type-1 transform
(modes on the lattice) instead of the general type-3 one.

Semantics of the reference's ``core/antenna_gridding.py`` (:6-219): pick the shortest non-zero
baseline and the shortest one not collinear with it as a 2-D basis, express every antenna offset
in that basis, and accept if a common integer factor <= ``max_factor`` makes all coordinates
integral (rational reconstruction with bounded denominators).  Host-side, once per simulation.
"""

from __future__ import annotations

from fractions import Fraction
from math import lcm
from typing import Any, Dict, Tuple

import numpy as np


def find_integer_multiplier(arr: np.ndarray, max_denominator: int = 10**6) -> int:
    """Smallest positive f with f * arr integral under rational approximation (zeros ignored);
    reference core/antenna_gridding.py:6-35."""
    f = 1
    for v in np.ravel(arr):
        if v != 0:
            f = lcm(f, Fraction(float(v)).limit_denominator(max_denominator).denominator)
    return f


def can_scale_to_int(arr, tol: float = 1e-9, max_denominator: int = 10**6, max_factor=None):
    """(ok, factor): does an integer factor turn ``arr`` into integers within ``tol``?
    reference :38-72."""
    f = find_integer_multiplier(arr, max_denominator)
    if max_factor is not None and f > max_factor:
        return False, f
    scaled = f * np.asarray(arr, dtype=float)
    return bool(np.allclose(scaled, np.round(scaled), atol=tol)), f


def find_lattice_basis(antpos: Dict[Any, np.ndarray], tol: float = 1e-9):
    """2x2 matrix whose columns are the two lattice vectors, or None when all antennas coincide;
    reference :74-137 (a collinear array gets (shortest baseline, e_y) stacked as rows, as there)."""
    xy = np.array([np.asarray(antpos[a], dtype=float)[:2] for a in antpos])
    diffs = (xy[:, None, :] - xy[None, :, :]).reshape(-1, 2)
    length = np.hypot(diffs[:, 0], diffs[:, 1])
    keep = length > tol
    if not keep.any():
        return None
    diffs = diffs[keep][np.argsort(length[keep], kind="stable")]
    b1 = diffs[0]
    cross = b1[0] * diffs[1:, 1] - b1[1] * diffs[1:, 0]
    ok = np.nonzero(np.abs(cross) > tol)[0]
    if ok.size == 0:
        return np.vstack([b1, np.array([0.0, 1.0])])
    return np.column_stack([b1, diffs[1 + ok[0]]])


def check_antpos_griddability(antpos: Dict[Any, np.ndarray], tol: float = 1e-9,
                              max_denominator: int = 10**6, max_factor: int = 1000
                              ) -> Tuple[bool, Dict[Any, np.ndarray], np.ndarray]:
    """(is_griddable, integer antenna coordinates, 3x3 basis matrix / factor); reference :139-219.
    Not griddable -> (False, antpos unchanged, identity)."""
    keys = list(antpos)
    vec = np.array([np.asarray(antpos[a], dtype=float) for a in keys])
    b2 = find_lattice_basis(antpos, tol=tol)
    if b2 is None:
        return False, antpos, np.eye(vec.shape[-1])
    basis = np.zeros((3, 3))
    basis[:2, :2] = b2
    basis[2, 2] = 1.0
    coords = np.linalg.solve(basis, (vec - vec[0]).T).T
    ok, factor = can_scale_to_int(coords.ravel(), tol=tol, max_denominator=max_denominator,
                                  max_factor=max_factor)
    if not ok:
        return False, antpos, np.eye(vec.shape[-1])
    grid = {a: np.round(factor * coords[i]).astype(int) for i, a in enumerate(keys)}
    return True, grid, basis / factor
