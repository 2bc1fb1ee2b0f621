"""Beam containers and the evaluator interface.

The reference takes pyuvdata ``UVBeam`` / ``BeamInterface`` / analytic beam objects
(src/fftvis/cpu/beams.py:12-89); pyuvdata is not available in this pipeline, so the package
carries two self-contained beam types that the GPU engine can put on the device, and duck-types
the corresponding pyuvdata objects when they are passed in (``describe_beam``).
"""

from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np


class BeamEvaluator(ABC):
    """Mirror of the reference's core/beams.py:10-93 (without the matvis base class)."""

    def __init__(self, **kwargs):
        self.beam_list = []
        self.beam_idx = None
        self.polarized = False
        self.freq = 0.0
        self.spline_opts = {}
        self.nsrc = 0

    @abstractmethod
    def evaluate_beam(self, beam, az, za, polarized, freq, check=False, spline_opts=None,
                      interpolation_function="az_za_map_coordinates"):
        """Beam response at (az, za, freq): (2, 2, nsrc) if polarized else (nsrc,)."""


def spline_order(spline_opts) -> int:
    """Interpolation order asked for by the reference's ``spline_opts``: ``{"order": n}`` for
    ``az_za_map_coordinates``, ``{"kx": n, "ky": n}`` for ``az_za_simple`` (RectBivariateSpline);
    linear (1) when absent -- the only values the reference's own tests pass
    (tests/test_cpu_beams.py:72,82,411,428)."""
    o = spline_opts or {}
    if "order" in o:
        return int(o["order"])
    return int(max(o.get("kx", 1), o.get("ky", 1)))


def checked_spline_order(spline_opts) -> int:
    """The orders the device interpolates: 1 (bilinear) and 3 (cubic B-spline, the reference CLI's
    default, cli.py:50,146); anything else fails loudly."""
    order = spline_order(spline_opts)
    if order not in (1, 3):
        raise NotImplementedError(f"GPU beam interpolation supports spline orders 1 and 3, not {order}")
    return order


class AiryBeam:
    """Analytic Airy dish: E-field 2 J1(x)/x, x = pi D nu sin(za)/c, in all four Jones slots;
    the power beam is its square (the pyuvdata ``AiryBeam`` used by the reference's tests,
    tests/test_beam_basis.py:33-42)."""

    def __init__(self, diameter: float):
        self.diameter = float(diameter)


class TabulatedBeam:
    """Beam sampled on a regular (za, az) grid, interpolated on the device (order 1 or 3).

    data : (nfreq_tab, 2, 2, nza, naz) complex E-field Jones [vector axis, feed]  -- or --
           (nfreq_tab, nza, naz) real power.  nfreq_tab is 1 (achromatic) or the number of
           simulated frequencies (tables already interpolated in frequency, which is what the
           reference's wrapper does up front, src/fftvis/wrapper.py:264-269).
    az is periodic: node j sits at 2 pi j / naz; za node i sits at za_max i / (nza - 1).
    """

    def __init__(self, data, freqs=None, za_max: float = np.pi):
        self.data = np.asarray(data)
        if self.data.ndim not in (3, 5):
            raise ValueError("TabulatedBeam data must be (nf, 2, 2, nza, naz) or (nf, nza, naz)")
        self.freqs = None if freqs is None else np.asarray(freqs, dtype=float)
        self.za_max = float(za_max)

    @property
    def is_efield(self) -> bool:
        return self.data.ndim == 5

    def power_from_efield(self, feed: int = 0) -> "TabulatedBeam":
        """Unpolarized power pattern of one feed, sum_ax |E[ax, feed]|^2 -- what matvis'
        ``prepare_beam_unpolarized`` hands the reference (src/fftvis/wrapper.py:278-279)."""
        if not self.is_efield:
            return self
        p = (np.abs(self.data[:, :, feed]) ** 2).sum(axis=1)
        return TabulatedBeam(p, self.freqs, self.za_max)


def describe_beam(beam, polarized: bool, freqs):
    """-> ("airy", diameter) or ("table", table ndarray, za_max) ready for the C ABI.
    ``freqs`` = the simulated frequencies (None: take the table's frequency axis as it is).

    Accepts this package's beams and duck-types pyuvdata's: a ``BeamInterface`` is unwrapped
    (``.beam``); an object with ``.diameter`` whose class name contains "Airy" is an Airy dish;
    a UVBeam-like object (``data_array`` (Naxes_vec, Nfeeds, Nfreqs, Nza, Naz), ``axis1_array``
    = az, ``axis2_array`` = za, regular axes, az starting at 0) becomes a table.
    """
    inner = getattr(beam, "beam", beam)
    if hasattr(inner, "diameter") and "airy" in type(inner).__name__.lower():
        return ("airy", float(inner.diameter))
    if isinstance(inner, TabulatedBeam):
        tb = inner
        if not polarized and tb.is_efield:
            tb = tb.power_from_efield()
        if polarized and not tb.is_efield:
            raise ValueError("polarized simulation needs an E-field beam table")
        data = tb.data
        if freqs is not None and data.shape[0] not in (1, len(freqs)):
            raise ValueError("beam table frequency axis must have length 1 or nfreqs")
        dt = np.complex128 if polarized else np.float64
        return ("table", np.ascontiguousarray(data, dtype=dt), tb.za_max)
    if hasattr(inner, "data_array") and hasattr(inner, "axis1_array"):
        az = np.asarray(inner.axis1_array, dtype=float)
        za = np.asarray(inner.axis2_array, dtype=float)
        d = np.asarray(inner.data_array)
        if d.ndim != 5:
            raise ValueError("UVBeam-like data_array must be (Naxes_vec, Nfeeds, Nfreqs, Nza, Naz)")
        if abs(az[0]) > 1e-12 or abs(za[0]) > 1e-12:
            raise ValueError("UVBeam-like axes must start at az = za = 0")
        if not np.isclose((az[1] - az[0]) * az.size, 2 * np.pi):
            raise ValueError("UVBeam-like azimuth axis must tile [0, 2 pi) periodically")
        tab = np.transpose(d, (2, 0, 1, 3, 4))
        bf = np.asarray(getattr(inner, "freq_array", freqs), dtype=float).ravel()
        if freqs is not None and tab.shape[0] > 1 and not (tab.shape[0] == len(freqs) and np.allclose(bf, freqs)):
            # linear interpolation in frequency, done once on the host (wrapper.py:264-269)
            idx = np.clip(np.searchsorted(bf, freqs) - 1, 0, bf.size - 2)
            wt = ((np.asarray(freqs) - bf[idx]) / (bf[idx + 1] - bf[idx]))[:, None, None, None, None]
            tab = tab[idx] * (1 - wt) + tab[idx + 1] * wt
        return describe_beam(TabulatedBeam(tab, freqs, float(za[-1])), polarized, freqs)
    raise NotImplementedError(
        f"beam of type {type(inner).__name__} cannot be placed on the GPU: pass an AiryBeam, a "
        "TabulatedBeam or an az/za UVBeam"
    )
