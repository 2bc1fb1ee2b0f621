"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the per-source astrometry the device applies from a per-time
context (fftvis_amd/csrc/fv_sim.h k_astrom_topo).

The reference delegates ICRS -> topocentric to matvis' ``CoordinateRotationERFA`` (cpu_simulate.py:693-709, rotated per
time at :937), which applies, per source, the published SOFA/ERFA chain: ``eraAtciqz`` (light deflection by the Sun,
annual aberration, bias-precession-nutation) and ``eraAtioq`` (Earth rotation, polar motion, diurnal aberration,
rotation to the horizon, refraction) under a per-time context ``eraASTROM`` (filled by ``eraApco13``).  Neither ERFA nor
matvis is in this pipeline: the formulas below are written from the published algorithm descriptions [MEM] -- PARITY
UNPINNED versus ERFA.  What they pin is the device kernel (same arithmetic, independent code) and its reduction to
the sidereal rotation for a trivial context.

A context is 31 float64 in eraASTROM's field order:
  pmt, eb[3], eh[3], em, v[3], bm1, bpn[9], along, phi, xpl, ypl, sphi, cphi, diurab, eral, refa, refb
"""

import numpy as np

SRS = 1.97412574336e-8  # Schwarzschild radius of the Sun, au
FIELDS = dict(pmt=0, eb=1, eh=4, em=7, v=8, bm1=11, bpn=12, along=21, phi=22, xpl=23, ypl=24, sphi=25, cphi=26,
              diurab=27, eral=28, refa=29, refb=30)


def icrs_to_enu(p, ctx):
    """(3, N) ICRS unit vectors -> (3, N) topocentric (east, north, up) unit vectors under one context."""
    p = np.asarray(p, dtype=float)
    c = np.asarray(ctx, dtype=float)
    eh, em, v, bm1 = c[4:7], c[7], c[8:11], c[11]
    bpn = c[12:21].reshape(3, 3)
    # light deflection by the Sun: p1 = p + w p x (e x q), q = p, w = SRS / em / max(q.(q + e), dlim)
    dlim = 1e-6 / max(em * em, 1.0)
    qdqpe = np.einsum("in,in->n", p, p + eh[:, None])
    w = SRS / em / np.maximum(qdqpe, dlim)
    eq = np.cross(eh[None, :], p.T).T
    p1 = p + w * np.cross(p.T, eq.T).T
    # annual aberration
    pdv = v @ p1
    w1 = 1.0 + pdv / (1.0 + bm1)
    w2 = SRS / em
    pa = p1 * bm1 + w1 * v[:, None] + w2 * (v[:, None] - pdv * p1)
    pa /= np.linalg.norm(pa, axis=0)
    # bias-precession-nutation -> CIRS, then Earth rotation (eral = ERA + longitude)
    ci = bpn @ pa
    se, ce = np.sin(c[28]), np.cos(c[28])
    x, y, z = ce * ci[0] + se * ci[1], -se * ci[0] + ce * ci[1], ci[2]
    # polar motion
    sx, cx, sy, cy = np.sin(c[23]), np.cos(c[23]), np.sin(c[24]), np.cos(c[24])
    xhd = cx * x + sx * z
    yhd = sx * sy * x + cy * y - cx * sy * z
    zhd = -sx * cy * x + sy * y + cx * cy * z
    # diurnal aberration
    f = 1.0 - c[27] * yhd
    xhdt, yhdt, zhdt = f * xhd, f * (yhd + c[27]), f * zhd
    # (-HA, Dec) -> (az, el) Cartesian: x towards south, y east, z up
    sphi, cphi = c[25], c[26]
    xaet = sphi * xhdt - cphi * zhdt
    yaet = yhdt
    zaet = cphi * xhdt + sphi * zhdt
    # refraction A tan z + B tan^3 z, guarded near the horizon
    r = np.maximum(np.hypot(xaet, yaet), 1e-6)
    zc = np.maximum(zaet, 0.05)
    tz = r / zc
    wr = c[30] * tz * tz
    dl = (c[29] + wr) * tz / (1.0 + (c[29] + 3.0 * wr) / (zc * zc))
    cosdel = 1.0 - dl * dl / 2.0
    fr = cosdel - dl * zc / r
    out = np.stack([yaet * fr, -xaet * fr, cosdel * zaet + dl * r])
    return out / np.linalg.norm(out, axis=0)


def sidereal_context(lst, lat):
    """The trivial context: no deflection / aberration / precession / polar motion / refraction, Earth rotation angle
    + longitude = the local sidereal angle ``lst``.  icrs_to_enu under it is the plain equatorial -> ENU rotation."""
    c = np.zeros(31)
    c[4:7] = [1.0, 0.0, 0.0]
    c[7] = 1e30
    c[11] = 1.0
    c[12:21] = np.eye(3).ravel()
    c[22] = lat
    c[25], c[26] = np.sin(lat), np.cos(lat)
    c[28] = lst
    return c


def plausible_context(seed, lat=-0.5362):
    """A context with every term switched on at realistic magnitudes (Earth's orbital velocity ~1e-4 c, Sun at ~1 au,
    a BPN rotation of ~0.3 degrees, arcsecond polar motion, diurnal aberration ~1.5e-6, sea-level refraction)."""
    rng = np.random.default_rng(seed)
    c = np.zeros(31)
    c[0] = 24.0 + rng.uniform(0, 1)
    c[1:4] = rng.normal(size=3)
    e = rng.normal(size=3)
    c[4:7] = e / np.linalg.norm(e)
    c[7] = rng.uniform(0.98, 1.02)
    v = rng.normal(size=3)
    c[8:11] = 0.99e-4 * v / np.linalg.norm(v)
    c[11] = np.sqrt(1.0 - c[8:11] @ c[8:11])
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    th = 5e-3
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    c[12:21] = (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)).ravel()
    c[21] = 0.374
    c[22] = lat
    c[23], c[24] = 1.1e-6, -0.8e-6
    c[25], c[26] = np.sin(lat), np.cos(lat)
    c[27] = 1.4e-6
    c[28] = rng.uniform(0, 2 * np.pi)
    c[29], c[30] = 2.8e-4, -3.0e-7
    return c
