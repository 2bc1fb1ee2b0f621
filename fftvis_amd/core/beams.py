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
    """The orders the device interpolates: everything scipy.ndimage.map_coordinates takes, 0 .. 5 (1 = bilinear
    and 3 = cubic B-spline, the reference CLI's default, cli.py:50,146, have unrolled kernels; 0, 2, 4, 5 share
    a general path); anything else fails loudly, as scipy does."""
    order = spline_order(spline_opts)
    if not 0 <= order <= 5:
        raise ValueError(f"spline order not supported: {order} (beam interpolation takes orders 0 .. 5)")
    return order


def bspline_weights(order: int, t):
    """Values of the order + 1 cardinal B-splines of degree ``order`` that are non-zero at position ``t`` in [0, 1]
    of a knot span (Cox - de Boor on uniform knots; the device's ``bspline_weights``, fv_sim.h)."""
    t = np.asarray(t, dtype=float)
    w = [np.ones_like(t)]
    for j in range(1, order + 1):
        saved = np.zeros_like(t)
        for r in range(j):
            tmp = w[r] / j
            w[r] = saved + (r + 1 - t) * tmp
            saved = (t + (j - r - 1)) * tmp
        w.append(saved)
    return w


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
    """Beam sampled on a regular (za, az) grid, interpolated on the device (orders 0 .. 5).

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


# Analytic third-party beams (objects with pyuvdata's ``compute_response``).  The reference evaluates them
# exactly at every source (cpu/beams.py:69-81); here they reach the device in one of two checked ways:
#  * closed form: an object with a ``diameter`` whose response at a set of probes equals THIS package's
#    Airy form 2 J1(x)/x times one complex factor per Jones slot to 1e-12 of the peak
#    (``fit_airy_closed_form``; pyuvdata's AiryBeam: 1/sqrt(2) in every slot) is evaluated in closed form
#    with those factors (fv_sim_set_beam_airy_scaled) -- probe and verify, never assumed;
#  * table: anything else is sampled through its own ``compute_response`` onto a regular (za, az) grid of
#    the visible hemisphere and interpolated on the device (cubic B-spline).  The node spacing is not
#    assumed either: ``sample_response`` measures the interpolation error of the table it has built against
#    ``compute_response`` at the points between the nodes (lowest and highest frequency) and halves the
#    spacing of the axis that misses the tolerance until it is met or the table would pass
#    FFTVIS_HIP_BEAM_TABLE_BYTES (default 2 GiB), in which case it raises.  An azimuth-independent response
#    (probed) is stored on 8 azimuth nodes instead of 720.
SAMPLED_START = {3: (181, 720), 1: (361, 1440)}  # first level: 0.5-degree (cubic) / 0.25-degree (linear) nodes
SAMPLED_AZ_SYMMETRIC_NODES = 8
CLOSED_FORM_TOL = 1e-12


def _table_bytes_limit() -> float:
    import os

    return float(os.environ.get("FFTVIS_HIP_BEAM_TABLE_BYTES", 2 * 2**30))


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


def _interp_table(tab, za_max, az, za, order):
    """Host twin of the device interpolant (fv_sim.h eval_jones / eval_power) for ONE (nza, naz) plane, used
    only to measure a sampled table's error: az periodic, za mirrored at both ends, order 1 bilinear, orders
    2 .. 5 interpolating B-splines (scipy's ``spline_filter1d`` makes the coefficients, as pyuvdata's
    az_za_map_coordinates -> scipy.ndimage.map_coordinates does for the reference, cpu/beams.py:69-74), 0 nearest."""
    nza, naz = tab.shape
    half = 0.0 if order & 1 else 0.5  # even orders: the centred B-spline's knots sit at half-integers
    fa = np.mod(az, 2 * np.pi) / (2 * np.pi / naz) + half
    fz = np.clip(za / (za_max / (nza - 1)), 0, nza - 1) + half
    ia = np.floor(fa).astype(int)
    iz = np.minimum(np.floor(fz).astype(int), nza - 2) if order & 1 else np.floor(fz).astype(int)
    ta, tz = fa - ia, fz - iz
    c = tab
    if order >= 2:
        from scipy.ndimage import spline_filter1d

        def coefs(x):
            return spline_filter1d(spline_filter1d(x, order=order, axis=0, mode="mirror"), order=order, axis=1, mode="grid-wrap")

        c = coefs(tab.real) + 1j * coefs(tab.imag) if np.iscomplexobj(tab) else coefs(np.asarray(tab, float))
    wa, wz, per = bspline_weights(order, ta), bspline_weights(order, tz), 2 * (nza - 1)
    out = 0
    for k in range(order + 1):
        jz = np.mod(iz - order // 2 + k, per)
        jz = np.where(jz < nza, jz, per - jz)
        for m in range(order + 1):
            out = out + c[jz, np.mod(ia - order // 2 + m, naz)] * (wz[k] * wa[m])
    return out


SAMPLED_PAD = 24  # za nodes past the horizon: the spline's end condition (mirror) is wrong for a pattern with
#                   a slope there, and its error decays by 2 - sqrt(3) per node: 24 nodes -> 2e-14


def _sample_planes(beam, polarized, freqs, use_feed, nza, naz):
    """Planes on ``nza`` nodes from the zenith to the horizon plus SAMPLED_PAD nodes beyond it (the object's own
    response there; where that is not finite, the odd reflection 2 f(horizon) - f(mirror node)).
    Returns (planes, za_max of the padded axis)."""
    h = 0.5 * np.pi / (nza - 1)
    za = h * np.arange(nza + SAMPLED_PAD)
    az = 2.0 * np.pi * np.arange(naz) / naz
    Z, A = np.meshgrid(za, az, indexing="ij")
    shape = (2, 2, nza + SAMPLED_PAD, naz) if polarized else (nza + SAMPLED_PAD, naz)
    out = np.stack([np.asarray(response_at(beam, polarized, f, A, Z, use_feed)).reshape(shape) for f in freqs])
    if not np.all(np.isfinite(out[..., nza:, :])):
        k = np.arange(1, SAMPLED_PAD + 1)
        out[..., nza - 1 + k, :] = 2 * out[..., nza - 1:nza, :] - out[..., nza - 1 - k, :]
    return out, float(za[-1])


def _table_error(beam, polarized, freqs2, tab2, za_max, use_feed, order, axis, rng):
    """Largest |table interpolant - compute_response| / peak over probes BETWEEN the nodes of ``axis``
    (0: za, 1: az; the other coordinate sits on nodes, so that the two axes are measured separately), at
    the frequencies ``freqs2`` the planes ``tab2`` were sampled at."""
    nzp, naz = tab2.shape[-2:]
    nza = nzp - SAMPLED_PAD  # nodes from the zenith to the horizon: the part the engine reads
    n = 512
    iz = np.concatenate([[0, 1, nza - 3, nza - 2], rng.integers(0, nza - 1, n)])  # both ends always
    ia = rng.integers(0, naz, iz.size)
    za = (iz + (0.5 if axis == 0 else 0.0)) * (0.5 * np.pi / (nza - 1))
    az = (ia + (0.5 if axis == 1 else 0.0)) * (2 * np.pi / naz)
    err, peak = 0.0, 0.0
    for f, plane in zip(freqs2, tab2):
        want = response_at(beam, polarized, f, az, za, use_feed)
        planes = plane.reshape(-1, nzp, naz)
        got = np.stack([_interp_table(pl, za_max, az, za, order) for pl in planes]).reshape(np.shape(want))
        err = max(err, float(np.max(np.abs(got - want))))
        peak = max(peak, float(np.max(np.abs(plane[..., :nza, :]))))
    return err / peak if peak > 0 else 0.0


def sample_response(beam, polarized: bool, freqs, use_feed: str = "x", order: int = 3, tol: float = 1e-7):
    """(nfreq, 2, 2, nza, naz) complex Jones or (nfreq, nza, naz) power table of an object that has
    pyuvdata's ``compute_response(az_array=, za_array=, freq_array=)`` (analytic beams,
    ``BeamInterface``), sampled from the zenith to just past the horizon -- the engine only ever looks above it
    (reference cpu/beams.py:69-81 calls the same method per slice; sampling it once is the table
    counterpart of wrapper.py:264-269's one-off frequency interpolation).  The node spacing is refined
    until the device interpolant reproduces ``compute_response`` between the nodes to ``tol`` of the peak
    (measured, see the note above SAMPLED_START); raises ValueError when no table within the byte limit does.
    Returns (table, za_max)."""
    freqs = np.atleast_1d(np.asarray(freqs, dtype=float))
    nza, naz = SAMPLED_START[3 if order >= 2 else 1]  # (the refinement below measures the order actually used)
    rng = np.random.default_rng(0)
    f2 = np.unique([freqs.min(), freqs.max()])
    # azimuth-independent response?  (probed at the top frequency on a coarse (za, az) lattice)
    pz = np.repeat(np.linspace(0.02, 0.5 * np.pi - 0.02, 16), 8)
    pa = np.tile(2 * np.pi * (np.arange(8) + 0.37) / 8, 16)
    pr = np.asarray(response_at(beam, polarized, f2[-1], pa, pz, use_feed))
    pr = pr.reshape(pr.shape[:-1] + (16, 8))
    if np.max(np.abs(pr - pr[..., :1])) <= 1e-13 * max(np.max(np.abs(pr)), 1e-300):
        naz = SAMPLED_AZ_SYMMETRIC_NODES
    cell = 64 if polarized else 8
    last = None
    while True:
        if freqs.size * (nza + SAMPLED_PAD) * naz * cell > _table_bytes_limit():
            raise ValueError(
                f"beam {type(getattr(beam, 'beam', beam)).__name__}: no (za, az) table within "
                f"{_table_bytes_limit() / 2**30:.1f} GiB interpolates compute_response to {tol:g} of its peak at "
                f"order {order} (reached {last}); use beam_spline_opts order 3, a closed-form beam, or raise "
                "FFTVIS_HIP_BEAM_TABLE_BYTES")
        tab2, za_max = _sample_planes(beam, polarized, f2, use_feed, nza, naz)
        ez = _table_error(beam, polarized, f2, tab2, za_max, use_feed, order, 0, rng)
        ea = 0.0 if naz == SAMPLED_AZ_SYMMETRIC_NODES else _table_error(beam, polarized, f2, tab2, za_max, use_feed, order, 1, rng)
        last = f"{max(ez, ea):.1e} with {nza} x {naz} nodes"
        if ez <= tol and ea <= tol:
            break
        if ez > tol:
            nza = 2 * (nza - 1) + 1
        if ea > tol:
            naz *= 2
    if freqs.size == f2.size and np.array_equal(freqs, f2):
        return tab2, za_max
    return _sample_planes(beam, polarized, freqs, use_feed, nza, naz)


def fit_airy_closed_form(beam, polarized: bool, freqs, use_feed: str = "x"):
    """If the object has a ``diameter`` and its ``compute_response`` equals 2 J1(x)/x,
    x = pi D nu sin(za)/c, times ONE complex factor per Jones slot (polarized) / one real factor on the
    squared pattern (power) at every one of 72 probes (24 directions x lowest, middle, highest frequency)
    to CLOSED_FORM_TOL of the peak: (diameter, jones_scale (2, 2) complex, power_scale).  Else None --
    nothing is assumed about an object from its name."""
    inner = getattr(beam, "beam", beam)
    D = getattr(inner, "diameter", None)
    if D is None or not np.isscalar(D) or not float(D) > 0:
        return None
    from scipy.special import j1

    freqs = np.atleast_1d(np.asarray(freqs, dtype=float))
    rng = np.random.default_rng(1)
    za = np.concatenate([[0.0, 1e-9], rng.uniform(0.0, 0.5 * np.pi, 22)])
    az = rng.uniform(0.0, 2 * np.pi, za.size)
    es, rs = [], []
    for f in np.unique([freqs.min(), freqs[freqs.size // 2], freqs.max()]):
        x = np.pi * float(D) * f * np.sin(za) / 299792458.0
        es.append(np.where(x == 0.0, 1.0, 2.0 * j1(x) / np.where(x == 0.0, 1.0, x)))
        rs.append(response_at(beam, polarized, f, az, za, use_feed))
    e = np.concatenate(es)
    r = np.concatenate(rs, axis=-1)
    basis = e if polarized else e * e
    scale = (r * basis).sum(axis=-1) / (basis * basis).sum()  # least squares, one factor per slot
    peak = np.max(np.abs(r))
    if not peak > 0 or np.max(np.abs(r - scale[..., None] * basis)) > CLOSED_FORM_TOL * peak:
        return None
    if polarized:
        return float(D), np.asarray(scale, dtype=complex).reshape(2, 2), 1.0
    return float(D), np.ones((2, 2), dtype=complex), float(np.real(scale))


def describe_beam(beam, polarized: bool, freqs, use_feed: str = "x", order: int = 1, tol: float = 1e-7):
    """-> ("airy", diameter[, jones_scale, power_scale]) or ("table", table ndarray, za_max) ready for the C ABI.
    ``freqs`` = the simulated frequencies (None: take the table's frequency axis as it is).

    Accepts this package's beams (``AiryBeam`` is evaluated in closed form, ``TabulatedBeam`` is
    uploaded) and duck-types pyuvdata's: a UVBeam-like object (``data_array`` (Naxes_vec, Nfeeds,
    Nfreqs, Nza, Naz), ``axis1_array`` = az, ``axis2_array`` = za, regular axes, az starting at 0),
    bare or inside a ``BeamInterface`` (``.beam``), becomes a table; any other object with a
    ``compute_response`` method (pyuvdata's analytic beams) is probed through that method: it runs in
    closed form only if the probes prove it is this package's Airy form up to constant factors
    (``fit_airy_closed_form``), else it is SAMPLED through the method (``sample_response``, spacing refined
    to ``tol``) -- the engine follows the object, it never substitutes a formula on the strength of a name.
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
        fit = fit_airy_closed_form(beam, polarized, freqs, use_feed)
        if fit is not None:  # verified at the probes: this package's Airy form times the fitted factors
            return ("airy",) + fit
        tab, za_max = sample_response(beam, polarized, freqs, use_feed, order, tol)
        return describe_beam(TabulatedBeam(tab, freqs, za_max), polarized, freqs)
    raise NotImplementedError(
        f"beam of type {type(inner).__name__} cannot be placed on the GPU: pass an AiryBeam, a "
        "TabulatedBeam, an az/za UVBeam or an object with pyuvdata's compute_response"
    )


def table_tolerance(eps: float) -> float:
    """Interpolation error (relative to the beam's peak) a sampled analytic beam's table is refined to: the
    run's NUFFT tolerance, kept between 1e-9 (a cubic table much finer than that is no longer small) and
    1e-5."""
    return float(min(1e-5, max(1e-9, eps)))


def airy_factors(desc) -> np.ndarray:
    """The 9 float64 of fv_sim_set_beam_airy_scaled / fv_beam_eval(kind 0): Jones factors (re, im) x 4, power
    factor -- all ones for this package's own AiryBeam."""
    js = np.ones((2, 2), dtype=complex) if len(desc) < 4 else np.asarray(desc[2], dtype=complex)
    ps = 1.0 if len(desc) < 4 else float(desc[3])
    return np.ascontiguousarray(np.concatenate([js.reshape(4).view(float), [ps]]))


def is_sampled_analytic(beam) -> bool:
    """True for the third-party analytic beams ``describe_beam`` samples through compute_response."""
    inner = getattr(beam, "beam", beam)
    if isinstance(inner, (AiryBeam, TabulatedBeam)) or hasattr(inner, "data_array"):
        return False
    return callable(getattr(beam, "compute_response", None)) or callable(getattr(inner, "compute_response", None))
