"""Eigenbeam (SVD) compression of a set of antenna beams -- host-side preprocessing.

Mirror of the reference's ``compute_beam_basis`` (src/fftvis/core/beam_basis.py:17-154): every
beam is sampled on one common (za, az) grid at a single frequency, the samples are stacked as
rows, and a thin SVD gives the basis beams (right singular vectors) and the per-beam coefficients
``U s`` (rows), truncated where ``s / s[0] < threshold`` (:140-145).  The result feeds the basis
path of the engine (``simulate_vis(beam=eigenbeams, beam_coefs=...)``,
cpu_simulate.py:303-470 -> fv_sim_set_basis).

The reference returns pyuvdata ``UVBeam`` copies; this package's device-ready container is
``TabulatedBeam``, so the eigenbeams come back as achromatic tables on the common grid.
"""

from __future__ import annotations

import numpy as np

from .beams import TabulatedBeam, describe_beam, is_sampled_analytic, response_at

_C = 299792458.0


def _common_grid(axis1_array, axis2_array, n_axis1, n_axis2, beams):
    """Azimuth nodes (periodic, 2 pi excluded) and zenith-angle nodes of the common grid
    (reference :101-117: the first az/za beam's own axes, else linspace defaults)."""
    if axis1_array is None:
        for b in beams:
            inner = getattr(b, "beam", b)
            if isinstance(inner, TabulatedBeam):
                nza, naz = inner.data.shape[-2:]
                return 2 * np.pi * np.arange(naz) / naz, np.linspace(0.0, inner.za_max, nza)
            if hasattr(inner, "axis1_array") and hasattr(inner, "axis2_array"):
                axis1_array, axis2_array = inner.axis1_array, inner.axis2_array
                break
        else:
            axis1_array = np.linspace(0.0, 2.0 * np.pi, n_axis1)
            axis2_array = np.linspace(0.0, np.pi, n_axis2)
    az = np.asarray(axis1_array, dtype=float)
    za = np.asarray(axis2_array, dtype=float)
    if az.ndim != 1 or za.ndim != 1 or az.size < 2 or za.size < 2:
        raise ValueError("axis1_array and axis2_array must be 1-D with at least two nodes.")
    if np.isclose(az[-1] - az[0], 2 * np.pi):  # closed azimuth axis: the last node repeats the first
        az = az[:-1]
    if abs(az[0]) > 1e-12 or abs(za[0]) > 1e-12:
        raise ValueError("the common grid must start at az = za = 0")
    if not (np.allclose(np.diff(az), az[1] - az[0]) and np.allclose(np.diff(za), za[1] - za[0])):
        raise ValueError("the common grid must be regular")
    if not np.isclose((az[1] - az[0]) * az.size, 2 * np.pi):
        raise ValueError("the azimuth axis must tile [0, 2 pi) periodically")
    return az, za


def _table_on_grid(tab, za_max, az, za):
    """Order-1 resampling of a (..., nza, naz) table (az periodic) onto the common grid -- the
    same interpolation the engine applies at source positions (fv_beam_eval)."""
    nza, naz = tab.shape[-2:]
    if naz == az.size and nza == za.size and np.isclose(za_max, za[-1]):
        return tab
    fa = az / (2 * np.pi) * naz
    ia = np.floor(fa).astype(int)
    ta = fa - ia
    ia0, ia1 = ia % naz, (ia + 1) % naz
    fz = np.clip(za / za_max, 0.0, 1.0) * (nza - 1)
    iz = np.minimum(np.floor(fz).astype(int), nza - 2)
    tz = (fz - iz)[:, None]
    lo = tab[..., iz, :]
    hi = tab[..., iz + 1, :]
    rowz = lo * (1 - tz) + hi * tz
    return rowz[..., ia0] * (1 - ta) + rowz[..., ia1] * ta


def _sample_beam(beam, freq, polarized, az, za):
    """One beam on the common grid: (2, 2, nza, naz) complex Jones, or (nza, naz) real power."""
    if is_sampled_analytic(beam):  # a third-party analytic beam answers for itself, exactly, on the grid
        Z, A = np.meshgrid(za, az, indexing="ij")
        r = response_at(beam, polarized, freq, A, Z)
        return r.reshape((2, 2) + Z.shape) if polarized else r.reshape(Z.shape)
    kind = describe_beam(beam, polarized, None)
    if kind[0] == "airy":
        from scipy.special import j1

        x = np.pi * kind[1] * freq * np.sin(za) / _C
        e = np.where(x == 0, 1.0, 2 * j1(x) / np.where(x == 0, 1.0, x))
        col = np.repeat(e[:, None], az.size, axis=1)
        if polarized:
            return np.broadcast_to(col.astype(complex), (2, 2) + col.shape).copy()
        return col * col
    _, tab, za_max = kind
    inner = getattr(beam, "beam", beam)
    bf = getattr(inner, "freqs", None)
    if bf is None:
        bf = getattr(inner, "freq_array", None)
    if tab.shape[0] == 1:
        sl = tab[0]
    else:
        if bf is None:
            raise ValueError("a chromatic beam table needs its frequency axis to be sampled at freq")
        bf = np.asarray(bf, dtype=float).ravel()
        i = int(np.clip(np.searchsorted(bf, freq) - 1, 0, bf.size - 2))
        t = (freq - bf[i]) / (bf[i + 1] - bf[i])
        sl = tab[i] * (1 - t) + tab[i + 1] * t
    return _table_on_grid(sl, za_max, az, za)


def compute_beam_basis(beam_list, freq: float, polarized: bool, threshold: float = 1e-12,
                       axis1_array=None, axis2_array=None, n_axis1: int = 361, n_axis2: int = 181):
    """SVD beam basis of ``beam_list`` at one frequency (reference core/beam_basis.py:17-154).

    Returns ``(eigenbeams, beam_coefs)``: a list of K ``TabulatedBeam`` (achromatic, on the
    common grid) and an ``(n_beams, K)`` array with ``beam_i = sum_k beam_coefs[i, k] eigenbeam_k``
    up to the discarded singular values.  Same argument meaning and error texts as the reference.
    """
    if len(beam_list) == 0:
        raise ValueError("beam_list must contain at least one beam.")
    if not (0.0 < threshold <= 1.0):
        raise ValueError("threshold must be in the interval (0, 1].")
    freq_grid = np.atleast_1d(freq).astype(float)
    if freq_grid.size != 1:
        raise ValueError("compute_beam_basis currently expects a scalar freq.")
    if (axis1_array is None) != (axis2_array is None):
        raise ValueError("axis1_array and axis2_array must be supplied together.")
    if polarized:
        for b in beam_list:
            inner = getattr(b, "beam", b)
            if isinstance(inner, TabulatedBeam) and not inner.is_efield:
                raise ValueError("polarized=True requires efield beams.")
            if getattr(inner, "beam_type", "efield") != "efield":
                raise ValueError("polarized=True requires efield beams.")

    az, za = _common_grid(axis1_array, axis2_array, n_axis1, n_axis2, beam_list)
    slices = [_sample_beam(b, float(freq_grid[0]), polarized, az, za) for b in beam_list]
    shape = slices[0].shape
    for idx, s in enumerate(slices):
        if s.shape != shape:
            raise ValueError(f"Beam {idx} evaluates to shape {s.shape}, expected {shape}.")
    flat = np.stack([s.ravel() for s in slices], axis=0)

    U, s, Vh = np.linalg.svd(flat, full_matrices=False)
    K = int(np.sum(s / s[0] >= threshold))
    beam_coefs = U[:, :K] * s[:K][None, :]
    eigenbeams = [TabulatedBeam(Vh[k].reshape((1,) + shape), None, float(za[-1])) for k in range(K)]
    return eigenbeams, beam_coefs


def compute_beam_basis_per_freq(beam_list, freqs, polarized: bool, nbasis: int,
                                axis1_array=None, axis2_array=None, n_axis1: int = 361,
                                n_axis2: int = 181):
    """Chromatic convenience on top of ``compute_beam_basis``: one SVD per simulated frequency,
    truncated to the same ``nbasis`` everywhere, returned in the layout the engine takes --
    ``nbasis`` tables with a frequency axis and ``beam_coefs`` of shape (n_beams, nbasis, nfreqs)
    (cpu_simulate.py:303-470 indexes ``beam_coefs[:, k, freq]``).  Singular-vector signs are fixed
    by making each basis beam's largest sample real-positive, so tables vary smoothly with
    frequency."""
    freqs = np.asarray(freqs, dtype=float)
    tabs, coefs = None, None
    for fi, f in enumerate(freqs):
        eb, c = compute_beam_basis(beam_list, float(f), polarized, 1e-300 if nbasis else 1.0,
                                   axis1_array, axis2_array, n_axis1, n_axis2)
        if len(eb) < nbasis:
            raise ValueError(f"only {len(eb)} basis beams exist at {f} Hz, {nbasis} requested")
        if tabs is None:
            shape = eb[0].data.shape[1:]
            tabs = np.zeros((nbasis, len(freqs)) + shape, dtype=eb[0].data.dtype)
            coefs = np.zeros((len(beam_list), nbasis, len(freqs)), dtype=c.dtype)
            za_max = eb[0].za_max
        for k in range(nbasis):
            d = eb[k].data[0]
            piv = d.ravel()[np.argmax(np.abs(d))]
            ph = np.abs(piv) / piv
            tabs[k, fi] = d * ph
            coefs[:, k, fi] = c[:, k] / ph
    return [TabulatedBeam(tabs[k], freqs, za_max) for k in range(nbasis)], coefs
