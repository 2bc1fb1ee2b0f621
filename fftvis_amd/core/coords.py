"""Source coordinates for the GPU engine.

The reference delegates ICRS -> topocentric astrometry to matvis' ``CoordinateRotationERFA``
(src/fftvis/cpu/cpu_simulate.py:693-704,937-946), which is not available in this pipeline.
The GPU engine consumes one 3x3 equatorial -> ENU rotation per time and applies it on the
device; ``SiderealRotation`` builds those matrices from the Julian date, longitude and latitude
(mean sidereal rotation only -- no precession / nutation / aberration / refraction; a documented
approximation).  A caller that owns a matvis coordinate manager can hand the
engine per-time topocentric unit vectors instead (``coord_mgr=`` in GPUSimulationEngine).
"""

from __future__ import annotations

import numpy as np


def latlon_of(telescope_loc):
    """(lat, lon) in radians from an astropy ``EarthLocation``-like object (``.lat/.lon`` with
    ``.rad``), an object with float ``.lat/.lon`` in radians, or a (lat, lon[, height]) tuple."""
    if hasattr(telescope_loc, "lat") and hasattr(telescope_loc, "lon"):
        lat, lon = telescope_loc.lat, telescope_loc.lon
        return float(getattr(lat, "rad", lat)), float(getattr(lon, "rad", lon))
    return float(telescope_loc[0]), float(telescope_loc[1])


def julian_dates(times) -> np.ndarray:
    """Julian dates from a numpy array or an astropy ``Time``-like object (``.jd``)."""
    if hasattr(times, "jd"):
        return np.atleast_1d(np.asarray(times.jd, dtype=float))
    return np.atleast_1d(np.asarray(times, dtype=float))


def gmst_rad(jd):
    """Greenwich mean sidereal time (IAU 1982 polynomial) in radians."""
    d = np.asarray(jd, dtype=float) - 2451545.0
    T = d / 36525.0
    deg = 280.46061837 + 360.98564736629 * d + 0.000387933 * T * T - T**3 / 38710000.0
    return np.deg2rad(np.mod(deg, 360.0))


def eq_unit_vectors(ra, dec) -> np.ndarray:
    """(3, N) equatorial unit vectors."""
    ra = np.asarray(ra, dtype=float)
    dec = np.asarray(dec, dtype=float)
    cd = np.cos(dec)
    return np.stack([cd * np.cos(ra), cd * np.sin(ra), np.sin(dec)])


class SiderealRotation:
    """Per-time equatorial -> (east, north, up) rotation matrices."""

    def __init__(self, times, telescope_loc):
        self.times = julian_dates(times)
        self.lat, self.lon = latlon_of(telescope_loc)

    def matrices(self) -> np.ndarray:
        lst = gmst_rad(self.times) + self.lon
        sl, cl = np.sin(lst), np.cos(lst)
        sp, cp = np.sin(self.lat), np.cos(self.lat)
        R = np.zeros((self.times.size, 3, 3))
        R[:, 0, 0], R[:, 0, 1] = -sl, cl
        R[:, 1, 0], R[:, 1, 1], R[:, 1, 2] = -sp * cl, -sp * sl, cp
        R[:, 2, 0], R[:, 2, 1], R[:, 2, 2] = cp * cl, cp * sl, sp
        return R


# ---- per-time astrometry contexts for the device-side coordinate manager (fv_sim_set_astrom) ---------------------
# 31 float64 per time, ERFA's eraASTROM in its field order:
#   pmt, eb[3], eh[3], em, v[3], bm1, bpn[9], along, phi, xpl, ypl, sphi, cphi, diurab, eral, refa, refb
ASTROM_LEN = 31


def sidereal_astrom_context(times, telescope_loc) -> np.ndarray:
    """(ntimes, 31) contexts under which the device-side astrometry reduces to ``SiderealRotation``: no deflection
    (Sun at 1e30 au), no aberration (v = 0), identity bias-precession-nutation, no polar motion / diurnal
    aberration / refraction, and eral = GMST + longitude."""
    jd = julian_dates(times)
    lat, lon = latlon_of(telescope_loc)
    c = np.zeros((jd.size, ASTROM_LEN))
    c[:, 4] = 1.0
    c[:, 7] = 1e30
    c[:, 11] = 1.0
    c[:, 12:21] = np.eye(3).ravel()
    c[:, 21], c[:, 22] = lon, lat
    c[:, 25], c[:, 26] = np.sin(lat), np.cos(lat)
    c[:, 28] = np.mod(gmst_rad(jd) + lon, 2 * np.pi)
    return c


def erfa_astrom_context(times, telescope_loc, pressure=0.0) -> np.ndarray:
    """(ntimes, 31) contexts from astropy / ERFA -- the astrometry context astropy's own ICRS -> AltAz
    transformation uses (``erfa_astrom.get().apco(AltAz(obstime, location, pressure))``, i.e. ``erfa.apco`` with the
    IERS Earth-orientation data astropy carries), one per time.  This is source-independent work of microseconds per
    time; the per-source part then runs on the device.  astropy is a dependency of the reference, not of this
    backend: imported here, on request only; raises ValueError without it."""
    try:
        from astropy import units as un
        from astropy.coordinates import AltAz, EarthLocation
        from astropy.coordinates.erfa_astrom import erfa_astrom
        from astropy.time import Time
    except ImportError as e:
        raise ValueError(f"device_astrometry=True needs astropy ({e}); pass astrom= contexts built elsewhere, or "
                         "coord_mgr=, or coord_method='SiderealRotation'") from e
    t = times if hasattr(times, "jd") else Time(np.asarray(times, dtype=float), format="jd")
    loc = telescope_loc
    if not isinstance(loc, EarthLocation):
        lat, lon = latlon_of(loc)
        height = float(loc[2]) if not hasattr(loc, "lat") and len(loc) > 2 else 0.0
        loc = EarthLocation.from_geodetic(lon * un.rad, lat * un.rad, height * un.m)
    out = np.zeros((len(t), ASTROM_LEN))
    for i in range(len(t)):
        a = erfa_astrom.get().apco(AltAz(obstime=t[i], location=loc, pressure=pressure * un.hPa))
        row = [a["pmt"], *np.ravel(a["eb"]), *np.ravel(a["eh"]), a["em"], *np.ravel(a["v"]), a["bm1"],
               *np.ravel(a["bpn"]), a["along"], a["phi"], a["xpl"], a["ypl"], a["sphi"], a["cphi"], a["diurab"],
               a["eral"], a["refa"], a["refb"]]
        out[i] = np.asarray(row, dtype=float).ravel()
    return out
