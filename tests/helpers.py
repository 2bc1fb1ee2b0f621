"""Shared helpers for the parity tests: oracle twins of the product's beam containers."""

import numpy as np

import fftvis_amd
from fftvis_amd.core.beams import spline_order
from oracle import fftvis_oracle as orc


def rel_l2(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / (nb if nb else 1.0)


def oracle_beam(beam, polarized, freqs, order=1, use_feed="x"):
    """The oracle's twin of a product beam.  Unpolarized runs go through the ORACLE's own restatement of
    ``prepare_beam_unpolarized`` (E-field -> power of one feed), never through the product's reduction; objects
    that are neither of this package's containers (third-party analytic beams) are handed over as they are: the
    oracle calls their ``compute_response`` at every source, as the reference does (cpu/beams.py:69-81)."""
    if isinstance(beam, fftvis_amd.AiryBeam):
        # this package's dish is DEFINED with both types (E-field e in every slot, power e^2): no reduction involved
        return orc.AiryBeam(beam.diameter, "efield" if polarized else "power")
    elif isinstance(beam, fftvis_amd.TabulatedBeam):
        ob = orc.TabulatedBeam(beam.data, freqs, beam.za_max, "efield" if beam.is_efield else "power", order)
    else:
        ob = beam
    return ob if polarized else orc.prepare_beam_unpolarized(ob, use_feed)


def oracle_simulate(cfg):
    """Run the oracle on simulate_vis-style keyword arguments."""
    beams = cfg["beam"] if isinstance(cfg["beam"], list) else [cfg["beam"]]
    order = spline_order(cfg.get("beam_spline_opts"))
    ob = [oracle_beam(b, cfg["polarized"], cfg["freqs"], order, cfg.get("use_feed", "x")) for b in beams]
    return orc.simulate(
        cfg["ants"], cfg["freqs"], cfg["fluxes"], ob, cfg["ra"], cfg["dec"], cfg["times"],
        cfg["telescope_loc"], baselines=cfg.get("baselines"), beam_idx=cfg.get("beam_idx"),
        polarized=cfg["polarized"], beam_coefs=cfg.get("beam_coefs"),
        force_use_type3=cfg.get("force_use_type3", True),
        reference_compat=cfg.get("reference_compat", True),
    )


def install_reference_dependency_stubs(monkeypatch):
    """Minimal stand-ins for the pieces of astropy / matvis the reference's engine touches when it builds its
    coordinate manager (cpu_simulate.py:686-709), put into ``sys.modules`` for one test: ``astropy.units``
    (``rad``, ``s``), ``astropy.time.Time(jd, format="jd")`` (indexing, differences with ``.to``),
    ``astropy.coordinates.SkyCoord`` and ``matvis.core.coords.CoordinateRotation`` with its ``_methods`` registry
    and one subclass, ``CoordinateRotationERFA``, whose vectors come from the oracle's sidereal stand-in.
    Returns the list that records every manager constructed (kwargs, ``_set_bcrs`` calls, rotations)."""
    import sys
    import types

    made = []

    class Unit:
        __array_ufunc__ = None  # ndarray * unit defers to __rmul__, as astropy's units do

        def __init__(self, name):
            self.name = name

        def __rmul__(self, other):
            return Quantity(np.asarray(other, dtype=float), self)

    class Quantity:
        def __init__(self, value, unit):
            self.value, self.unit = value, unit

        def to(self, unit):
            scale = {("d", "s"): 86400.0, ("s", "s"): 1.0, ("rad", "rad"): 1.0}[(self.unit.name, unit.name)]
            return Quantity(self.value * scale, unit)

        def __lt__(self, other):
            return self.value < (other.value if isinstance(other, Quantity) else other)

        def __gt__(self, other):
            return self.value > (other.value if isinstance(other, Quantity) else other)

    un = types.ModuleType("astropy.units")
    un.rad, un.s, un.d = Unit("rad"), Unit("s"), Unit("d")

    class Time:
        def __init__(self, val, format="jd"):
            assert format == "jd"
            self.jd = np.asarray(val, dtype=float)

        def __len__(self):
            return self.jd.size

        def __getitem__(self, i):
            return Time(self.jd[i])

        def __sub__(self, other):
            return Quantity(self.jd - other.jd, un.d)

    class SkyCoord:
        def __init__(self, ra, dec, frame):
            assert frame == "icrs" and ra.unit.name == "rad" and dec.unit.name == "rad"
            self.ra, self.dec = ra, dec

    class CoordinateRotation:
        _methods = {}

        def __init_subclass__(cls):
            CoordinateRotation._methods[cls.__name__] = cls

    class CoordinateRotationERFA(CoordinateRotation):
        def __init__(self, flux, times, telescope_loc, skycoords, chunk_size=None, source_buffer=1.0, precision=1,
                     update_bcrs_every=0.0):
            self.kw = dict(flux=flux, times=times, telescope_loc=telescope_loc, skycoords=skycoords,
                           chunk_size=chunk_size, source_buffer=source_buffer, precision=precision)
            self.update_bcrs_every = update_bcrs_every
            self.times = times
            self.bcrs_set, self.rotated, self.setup_calls = [], [], 0
            self.o = orc.SimpleCoordinateRotation(flux, times.jd, telescope_loc, skycoords.ra.value,
                                                  skycoords.dec.value)
            made.append(self)

        def _set_bcrs(self, t):
            self.bcrs_set.append(t)

        def setup(self):
            self.setup_calls += 1

        def rotate(self, ti):
            self.rotated.append(ti)
            self.o.rotate(ti)
            self.all_coords_topo = self.o._topo

    mods = {
        "astropy": types.ModuleType("astropy"), "astropy.units": un,
        "astropy.time": types.ModuleType("astropy.time"),
        "astropy.coordinates": types.ModuleType("astropy.coordinates"),
        "matvis": types.ModuleType("matvis"), "matvis.core": types.ModuleType("matvis.core"),
        "matvis.core.coords": types.ModuleType("matvis.core.coords"),
    }
    mods["astropy"].units = un
    mods["astropy.time"].Time = Time
    mods["astropy.coordinates"].SkyCoord = SkyCoord
    mods["matvis.core.coords"].CoordinateRotation = CoordinateRotation
    mods["matvis"].core = mods["matvis.core"]
    mods["matvis.core"].coords = mods["matvis.core.coords"]
    for name, m in mods.items():
        monkeypatch.setitem(sys.modules, name, m)
    return made, Time
