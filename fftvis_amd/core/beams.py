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
    """Analytic Airy dish evaluated in closed form on the device.

    THIS PACKAGE'S definition, not pyuvdata's: E-field ``2 J1(x)/x``, ``x = pi D nu sin(za)/c``, in
    all four Jones slots ``A[ax, feed]`` (so a polarized simulation of an unpolarized sky gives
    ``V_xx = 2 x`` the unpolarized result); the power beam is its square.  Third-party analytic beams
    (pyuvdata's ``AiryBeam`` and friends, whose Jones normalisation is theirs to define) are never
    mapped onto this formula: ``describe_beam`` samples their own ``compute_response`` instead."""

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

    def power_from_efield(self, feed: int = 0) -> "TabulatedBeam":  # feed: 0 = "x", 1 = "y"
        """Unpolarized power pattern of one feed, sum_ax |E[ax, feed]|^2 -- what matvis'
        ``prepare_beam_unpolarized`` hands the reference (src/fftvis/wrapper.py:278-279)."""
        if not self.is_efield:
            return self
        p = (np.abs(self.data[:, :, feed]) ** 2).sum(axis=1)
        return TabulatedBeam(p, self.freqs, self.za_max)


FEED_ALIASES = {"x": ("x", "e"), "y": ("y", "n")}  # pyuvdata feed_array spellings


def feed_index(use_feed: str, feed_array=None) -> int:
    """Index of the feed ``use_feed`` ("x" or "y", reference wrapper.py:101 / matvis
    ``prepare_beam_unpolarized(beam, use_feed=...)``) in a beam's feed axis: looked up in its
    ``feed_array`` when it has one, else x -> 0, y -> 1.  Anything else raises."""
    key = str(use_feed).lower()
    if key not in FEED_ALIASES:
        raise ValueError(f"use_feed must be 'x' or 'y', not {use_feed!r}")
    if feed_array is not None:
        names = [str(f).lower() for f in np.asarray(feed_array).ravel()]
        for alias in FEED_ALIASES[key]:
            if alias in names:
                return names.index(alias)
        raise ValueError(f"beam has no feed {use_feed!r} (feed_array = {names})")
    return 0 if key == "x" else 1


# Analytic third-party beams are sampled onto a regular grid of the visible hemisphere and then
# interpolated on the device like any table: cubic B-spline on 0.5-degree nodes follows a 14 m dish at
# 250 MHz (pattern scale ~1.6 degrees) to ~1e-5 of its peak; order 1 needs the finer grid for ~2e-3.
SAMPLED_NODES = {3: (181, 720), 1: (361, 1440)}


def _response_object(beam):
    inner = getattr(beam, "beam", beam)
    return beam if callable(getattr(beam, "compute_response", None)) else inner


def response_at(beam, polarized: bool, freq: float, az, za, use_feed: str = "x"):
    """An object's own ``compute_response`` at flat (az, za) arrays and one frequency, reduced to what the
    reference's evaluate_beam keeps (cpu/beams.py:76-81): (2, 2, npts) complex Jones [vector axis, feed]
    when polarized, else (npts,) real power of one feed / polarisation."""
    obj = _response_object(beam)
    inner = getattr(obj, "beam", obj)
    btype = str(getattr(obj, "beam_type", getattr(inner, "beam_type", "efield"))).lower()
    az = np.ascontiguousarray(az, dtype=float).ravel()
    za = np.ascontiguousarray(za, dtype=float).ravel()
    r = np.asarray(obj.compute_response(az_array=az, za_array=za, freq_array=np.atleast_1d(float(freq))))
    if r.ndim != 4 or r.shape[-1] != az.size:
        raise ValueError(f"compute_response returned shape {r.shape}; expected (Naxes_vec, Nfeeds, 1, Npts)")
    r = r[:, :, 0, :]
    if polarized:
        if btype == "power" or r.shape[:2] != (2, 2):
            raise ValueError("polarized simulation needs an E-field beam with 2 vector axes and 2 feeds")
        return r.astype(complex)
    if btype == "power":  # (1, Npols, Npts): one polarisation, as prepare_beam_unpolarized leaves it
        return np.real(r[0, 0 if r.shape[1] == 1 else feed_index(use_feed)])
    k = 0 if r.shape[1] == 1 else feed_index(use_feed, getattr(inner, "feed_array", None))
    return (np.abs(r[:, k]) ** 2).sum(axis=0)  # power of one feed = sum over the vector axes of |E|^2


def sample_response(beam, polarized: bool, freqs, use_feed: str = "x", order: int = 3):
    """(nfreq, 2, 2, nza, naz) complex Jones or (nfreq, nza, naz) power table of an object that has
    pyuvdata's ``compute_response(az_array=, za_array=, freq_array=)`` (analytic beams,
    ``BeamInterface``), sampled for za in [0, pi/2] -- the engine only ever looks above the horizon
    (reference cpu/beams.py:69-81 calls the same method per slice; sampling it once is the table
    counterpart of wrapper.py:264-269's one-off frequency interpolation).  Returns (table, za_max)."""
    nza, naz = SAMPLED_NODES[3 if order == 3 else 1]
    za = np.linspace(0.0, 0.5 * np.pi, nza)
    az = 2.0 * np.pi * np.arange(naz) / naz
    Z, A = np.meshgrid(za, az, indexing="ij")
    shape = (2, 2, nza, naz) if polarized else (nza, naz)
    out = [response_at(beam, polarized, f, A, Z, use_feed).reshape(shape)
           for f in np.atleast_1d(np.asarray(freqs, dtype=float))]
    return np.stack(out), 0.5 * np.pi


def describe_beam(beam, polarized: bool, freqs, use_feed: str = "x", order: int = 1):
    """-> ("airy", diameter) or ("table", table ndarray, za_max) ready for the C ABI.
    ``freqs`` = the simulated frequencies (None: take the table's frequency axis as it is).

    Accepts this package's beams (``AiryBeam`` is evaluated in closed form, ``TabulatedBeam`` is
    uploaded) and duck-types pyuvdata's: a UVBeam-like object (``data_array`` (Naxes_vec, Nfeeds,
    Nfreqs, Nza, Naz), ``axis1_array`` = az, ``axis2_array`` = za, regular axes, az starting at 0),
    bare or inside a ``BeamInterface`` (``.beam``), becomes a table; any other object with a
    ``compute_response`` method (pyuvdata's analytic beams) is SAMPLED through that method
    (``sample_response``) -- the engine follows the object, it never substitutes a formula of its own.
    Unpolarized runs take the power of feed ``use_feed`` of an E-field beam (reference
    wrapper.py:278-279).
    """
    inner = getattr(beam, "beam", beam)
    if isinstance(inner, AiryBeam):
        return ("airy", float(inner.diameter))
    if isinstance(inner, TabulatedBeam):
        tb = inner
        if not polarized and tb.is_efield:
            tb = tb.power_from_efield(feed_index(use_feed))
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
        if str(getattr(inner, "beam_type", "efield")).lower() == "power":
            if polarized:
                raise ValueError("polarized simulation needs an E-field beam, not a power beam")
            k = 0 if tab.shape[2] == 1 else feed_index(use_feed)  # (nf, 1, Npols, nza, naz)
            return describe_beam(TabulatedBeam(np.real(tab[:, 0, k]), freqs, float(za[-1])), False, freqs)
        if not polarized:  # the named feed, looked up in the object's own feed_array
            k = feed_index(use_feed, getattr(inner, "feed_array", None))
            return describe_beam(TabulatedBeam(tab, freqs, float(za[-1])).power_from_efield(k), False, freqs)
        return describe_beam(TabulatedBeam(tab, freqs, float(za[-1])), polarized, freqs)
    if callable(getattr(beam, "compute_response", None)) or callable(getattr(inner, "compute_response", None)):
        if freqs is None:
            raise ValueError("sampling an analytic beam needs the simulated frequencies")
        tab, za_max = sample_response(beam, polarized, freqs, use_feed, order)
        return describe_beam(TabulatedBeam(tab, freqs, za_max), polarized, freqs)
    raise NotImplementedError(
        f"beam of type {type(inner).__name__} cannot be placed on the GPU: pass an AiryBeam, a "
        "TabulatedBeam, an az/za UVBeam or an object with pyuvdata's compute_response"
    )


def is_sampled_analytic(beam) -> bool:
    """True for the third-party analytic beams ``describe_beam`` samples through compute_response."""
    inner = getattr(beam, "beam", beam)
    if isinstance(inner, (AiryBeam, TabulatedBeam)) or hasattr(inner, "data_array"):
        return False
    return callable(getattr(beam, "compute_response", None)) or callable(getattr(inner, "compute_response", None))
