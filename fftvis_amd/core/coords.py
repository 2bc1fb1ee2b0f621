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
