"""Seeded synthetic inputs for the benchmark configurations (BASELINE.json ``configs``,
SURVEY.md section 8(d)).  Pure numpy; used by bench.py, __graft_entry__.smoke() and the tests.

The antenna layouts are documented-synthetic stand-ins for ``hera_sim.antpos.hex_array``
(reference docs/tutorials/fftvis_gridded_array.ipynb:107), which is not importable here.
"""

from __future__ import annotations

import numpy as np

from .core.beams import AiryBeam, TabulatedBeam

HERA_LAT = np.deg2rad(-30.7215)  # reference tests/test_wrapper.py:81-85
HERA_LON = np.deg2rad(21.4283)
SPACING = 14.6


def hex_positions(side: int, spacing: float = SPACING) -> np.ndarray:
    """Filled hexagon with ``side`` antennas per edge: 3 side (side - 1) + 1 positions, z = 0."""
    pts = []
    for q in range(-(side - 1), side):
        for r in range(max(-(side - 1), -q - (side - 1)), min(side - 1, -q + side - 1) + 1):
            pts.append((spacing * (q + 0.5 * r), spacing * (np.sqrt(3) / 2) * r, 0.0))
    pts = np.array(pts)
    order = np.lexsort((pts[:, 0], pts[:, 1]))
    return pts[order]


def hera_like_array(kind: str) -> dict:
    """``hera7`` (side 2), ``hera37`` (side 4), ``hera350``: a split-core side-11 hexagon
    (three 120-degree sectors displaced by thirds of the lattice vectors, one seam row removed:
    320 antennas) plus 30 outriggers on the fifth ring of a 6x-spacing lattice (longest baseline
    ~876 m)."""
    if kind == "hera7":
        pos = hex_positions(2)
    elif kind == "hera37":
        pos = hex_positions(4)
    elif kind == "hera350":
        core = hex_positions(11)
        ang = np.mod(np.arctan2(core[:, 1], core[:, 0]), 2 * np.pi)
        r = np.hypot(core[:, 0], core[:, 1])
        seam = (np.abs(ang) < 1e-9) | (r < 1e-9)  # centre + the 10 antennas due east of it
        core, ang = core[~seam], ang[~seam]
        sector = (ang // (2 * np.pi / 3)).astype(int)
        a1 = np.array([SPACING, 0.0, 0.0])
        a2 = np.array([SPACING / 2, SPACING * np.sqrt(3) / 2, 0.0])
        shift = np.array([0 * a1, (a1 + a2) / 3.0, (2 * a2 - a1) / 3.0])
        core = core + shift[sector]
        ring = []
        big = 6 * SPACING
        n = 5
        for q in range(-n, n + 1):
            for rr in range(max(-n, -q - n), min(n, -q + n) + 1):
                if max(abs(q), abs(rr), abs(q + rr)) == n:
                    ring.append((big * (q + 0.5 * rr), big * (np.sqrt(3) / 2) * rr, 0.0))
        pos = np.vstack([core, np.array(ring)])
        assert len(pos) == 350
    elif kind == "scattered350":
        # 350 antennas drawn uniformly inside the convex hull of hera350 (seeded): the same extent and
        # baseline count, NO repeated baseline vectors and no lattice -- the generic type-3 workload
        # (finufft's nufft2d3 contract has no lattice in it, reference cpu/nufft.py:48-59)
        ref = np.array(list(hera_like_array("hera350").values()))
        from scipy.spatial import Delaunay

        hull = Delaunay(ref[:, :2])
        rng = np.random.default_rng(350)
        lo, hi = ref[:, :2].min(0), ref[:, :2].max(0)
        pts = np.empty((0, 2))
        while len(pts) < 350:
            cand = rng.uniform(lo, hi, size=(1024, 2))
            pts = np.vstack([pts, cand[hull.find_simplex(cand) >= 0]])
        pos = np.column_stack([pts[:350], np.zeros(350)])
    else:
        raise ValueError(kind)
    return {i: p for i, p in enumerate(pos)}


def with_z_scatter(ants: dict, sigma_m: float = 0.03, seed: int = 7) -> dict:
    """The same array with a seeded Gaussian height scatter (metres) about its plane: |b_z| then exceeds
    the reference's flat_array_tol = 1e-6 m (cpu_simulate.py:655), so the run takes the 3-D transform
    (cpu/nufft.py:62-118) -- every surveyed real array does."""
    rng = np.random.default_rng(seed)
    dz = rng.normal(0.0, sigma_m, len(ants))
    return {k: np.array([p[0], p[1], p[2] + dz[i]]) for i, (k, p) in enumerate(ants.items())}


def all_cross_baselines(ants: dict):
    keys = list(ants)
    return [(a, b) for i, a in enumerate(keys) for b in keys[i + 1:]]


def catalog(nsrc: int, freqs: np.ndarray, seed: int = 0, polarized_sky: bool = False):
    """Isotropic point sources; flux U(0,1) (nu/nu0)^-0.8 (reference fftvis_tutorial.ipynb
    cell 15); polarized sky adds Q,U,V ~ 0.1 I N(0,1)."""
    rng = np.random.default_rng(seed)
    ra = rng.uniform(0, 2 * np.pi, nsrc)
    dec = np.arcsin(rng.uniform(-1, 1, nsrc))
    amp = rng.uniform(0, 1, nsrc)
    flux = amp[:, None] * (np.asarray(freqs)[None, :] / freqs[0]) ** -0.8
    if polarized_sky:
        quv = 0.1 * flux[:, :, None] * rng.normal(size=(nsrc, 1, 3))
        flux = np.concatenate([flux[:, :, None], quv], axis=2)
    return ra, dec, flux


def synthetic_efield_table(freqs, diameter: float = 14.0, nza: int = 181, naz: int = 360):
    """Tabulated Jones [freq, vector axis, feed, za, az] on a 1-degree grid: Airy amplitude times
    a dipole-like rotation plus a smooth complex leakage term."""
    from scipy.special import j1

    za = np.linspace(0, np.pi, nza)
    az = 2 * np.pi * np.arange(naz) / naz
    Z, A = np.meshgrid(za, az, indexing="ij")
    tab = np.empty((len(freqs), 2, 2, nza, naz), dtype=complex)
    for i, f in enumerate(freqs):
        x = np.pi * diameter * f * np.sin(Z) / 299792458.0
        e = np.where(x == 0, 1.0, 2 * j1(x) / np.where(x == 0, 1, x))
        leak = 0.05 * (1 + 0.5j) * np.sin(Z)
        tab[i, 0, 0] = e * np.cos(A) + leak * np.sin(2 * A)
        tab[i, 0, 1] = e * np.sin(A) + leak * np.cos(A)
        tab[i, 1, 0] = -e * np.sin(A) * np.cos(Z) + leak * np.cos(2 * A)
        tab[i, 1, 1] = e * np.cos(A) * np.cos(Z) - leak * np.sin(A)
    return tab


CONFIGS = {
    # name: (array, nsrc, nfreq, ntimes, polarized, beam)
    "C1": ("hera7", 100, 8, 2, False, "airy"),
    "C2": ("hera37", 10_000, 64, 10, False, "airy"),
    "C3": ("hera350", 100_000, 128, 20, True, "table"),
    "C4": ("hera350", 1_000_000, 256, 60, True, "table"),
    # per-antenna beams through K = 4 basis beams (eigenbeam path), fp32, eps 1e-4
    "C5": ("hera350", 100_000, 128, 20, True, "basis"),
}
C5_NBASIS = 4


def make_config(name: str, seed: int = 0, nsrc=None, nfreq=None, ntimes=None, array=None, z_scatter: float = 0.0):
    """Inputs of one BASELINE.json configuration as a dict of simulate_vis keyword arguments
    (optionally shrunk for tests).  ``array`` replaces the configuration's layout (``"scattered350"``:
    no lattice, no repeated baseline vectors); ``z_scatter`` > 0 adds a seeded height scatter of that many
    metres (non-coplanar: the 3-D transform)."""
    arr, ns, nf, nt, pol, beamkind = CONFIGS[name]
    ns, nf, nt = nsrc or ns, nfreq or nf, ntimes or nt
    ants = hera_like_array(array or arr)
    if z_scatter > 0:
        ants = with_z_scatter(ants, z_scatter)
    freqs = np.linspace(100e6, 200e6, nf)
    times = np.linspace(2459845.0, 2459845.05, nt)
    ra, dec, flux = catalog(ns, freqs, seed)
    extra = {}
    if beamkind == "airy":
        beam = AiryBeam(14.0)
    elif beamkind == "table":
        beam = TabulatedBeam(synthetic_efield_table(freqs), freqs)
    else:
        # K basis tables (dishes of different diameter from the same generator) and per-antenna
        # coefficients: a dominant first mode plus a few-percent admixture of the others, the
        # structure the SVD of a +-7 % diameter scatter gives (beam_decomposition.ipynb cell 8)
        rng = np.random.default_rng(seed + 5)
        diam = 14.0 * (1 + 0.07 * np.linspace(-1, 1, C5_NBASIS))
        # real-valued tables: the eigenbeam path's V_lk = V_kl^T shortcut is exact only for those
        # (reference cpu_simulate.py:464-468)
        beam = [TabulatedBeam(synthetic_efield_table(freqs, d, nza=91, naz=180).real.astype(complex), freqs)
                for d in diam]
        nant = len(ants)
        coefs = 0.05 * (rng.normal(size=(nant, C5_NBASIS, nf)) + 1j * rng.normal(size=(nant, C5_NBASIS, nf)))
        coefs[:, 0, :] += 1.0
        extra = dict(beam_coefs=coefs, precision=1, eps=1e-4)
    cfg = dict(
        ants=ants, fluxes=flux, ra=ra, dec=dec, freqs=freqs, times=times, beam=beam,
        telescope_loc=(HERA_LAT, HERA_LON), baselines=all_cross_baselines(ants), polarized=pol,
        precision=2, eps=6e-8, force_use_type3=True,  # the benchmark path is the type-3 NUFFT
        # the documented stand-in for matvis/ERFA astrometry, by name (the engine refuses to fall back silently)
        coord_method="SiderealRotation",
    )
    cfg.update(extra)
    return cfg
