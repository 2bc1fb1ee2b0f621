"""GPU beam evaluator: the filled-in ``GPUBeamEvaluator`` of the reference
(stub at src/fftvis/gpu/beams.py:15-88; CPU twin src/fftvis/cpu/beams.py:9-246)."""

from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .. import _lib
from ..core.beams import BeamEvaluator, airy_factors, checked_spline_order, describe_beam, is_sampled_analytic
from ..core.utils import prepare_beam_evaluation as _prepare_beam_evaluation


def _coherency(variant, beam_i, beam_j, flux, device=0):
    bi = np.asarray(beam_i)
    prec = 1 if bi.dtype in (np.complex64, np.float32) else 2
    cdt = np.complex64 if prec == 1 else np.complex128
    rdt = np.float32 if prec == 1 else np.float64
    bi = np.ascontiguousarray(bi, dtype=cdt)
    bj = np.ascontiguousarray(beam_j, dtype=cdt)
    fl = np.ascontiguousarray(flux, dtype=cdt if variant in (1, 3) else rdt)
    n = bi.shape[-1]
    out = np.empty_like(bi)
    _lib.require_gpu()
    _lib.check(_lib.lib().fv_apparent_coherency(device, prec, variant, n, _lib.ptr(bi),
                                                _lib.ptr(bj), _lib.ptr(fl), _lib.ptr(out)))
    return out


class GPUBeamEvaluator(BeamEvaluator):
    """Beam evaluation and per-source coherency products on the MI355X."""

    def __init__(self, device: int = 0, **kwargs):
        super().__init__(**kwargs)
        self.device = device

    def evaluate_beam(self, beam, az: np.ndarray, za: np.ndarray, polarized: bool, freq: float,
                      check: bool = False, spline_opts: Optional[Dict] = None,
                      interpolation_function: str = "az_za_map_coordinates",
                      freq_index: int = 0) -> np.ndarray:
        """Beam at (az, za, freq): (2, 2, nsrc) [vector axis, feed, source] if ``polarized``
        else (nsrc,) power -- the shapes of reference cpu/beams.py:76-81.  ``freq_index``
        selects the plane of a multi-frequency table."""
        self.polarized = polarized
        self.freq = freq
        self.spline_opts = spline_opts or {}
        order = checked_spline_order(self.spline_opts)
        az = np.asarray(az)
        prec = 1 if az.dtype == np.float32 else 2
        rdt, cdt = (np.float32, np.complex64) if prec == 1 else (np.float64, np.complex128)
        az = np.ascontiguousarray(az, dtype=rdt)
        za = np.ascontiguousarray(za, dtype=rdt)
        n = az.size
        if is_sampled_analytic(beam):  # third-party analytic beam: closed form if its own response proves it is
            # this package's Airy form x constant factors, else a one-frequency table of its own response
            desc = describe_beam(beam, polarized, np.atleast_1d(float(freq)), order=3,
                                 tol=1e-5 if prec == 1 else 1e-9)
            order, freq_index = 3, 0
        else:
            desc = describe_beam(beam, polarized, None)
        out = np.empty((2, 2, n) if polarized else (n,), dtype=cdt)
        L = _lib.lib()
        _lib.require_gpu()
        if desc[0] == "airy":
            fac = airy_factors(desc)
            st = L.fv_beam_eval(self.device, prec, int(polarized), 0, desc[1], 0, 0, 0, 0.0, _lib.ptr(fac),
                                1, 0, float(freq), n, _lib.ptr(az), _lib.ptr(za), _lib.ptr(out))
        else:
            tab = desc[1]
            st = L.fv_beam_eval(self.device, prec, int(polarized), 1, 0.0, tab.shape[0],
                                tab.shape[-2], tab.shape[-1], float(desc[2]), _lib.ptr(tab), order,
                                int(freq_index), float(freq), n, _lib.ptr(az), _lib.ptr(za),
                                _lib.ptr(out))
        _lib.check(st)
        if check:  # reference cpu/beams.py:84-87
            sm = np.sum(out)
            if np.isinf(sm) or np.isnan(sm):
                raise ValueError("Beam interpolation resulted in an invalid value")
        return out

    prepare_beam_evaluation = staticmethod(_prepare_beam_evaluation)

    # The four per-source 2x2 products of reference cpu/beams.py:129-246.  Like their numba
    # twins the single-beam forms overwrite ``beam`` and the pair forms fill ``out``; all
    # return the result as well.
    def get_apparent_flux_polarized_beam(self, beam: np.ndarray, flux: np.ndarray):
        beam[...] = _coherency(0, beam, beam, flux, self.device)
        return beam

    def get_apparent_flux_polarized(self, beam: np.ndarray, flux: np.ndarray):
        """``flux`` (nsrc,) real -> (A^H A) I; ``flux`` (2, 2, nsrc) -> A^H C A
        (the stub's single entry point, reference gpu/beams.py:68-88)."""
        variant = 0 if np.ndim(flux) == 1 else 1
        beam[...] = _coherency(variant, beam, beam, flux, self.device)
        return beam

    def get_apparent_flux_polarized_beam_pair(self, beam_i, beam_j, flux, out):
        out[...] = _coherency(2, beam_i, beam_j, flux, self.device)
        return out

    def get_apparent_flux_polarized_pair(self, beam_i, beam_j, coherency, out):
        out[...] = _coherency(3, beam_i, beam_j, coherency, self.device)
        return out

    def get_apparent_flux_unpolarized(self, beam_i, beam_j, flux):
        """sqrt(B_i B_j) I (reference cpu_simulate.py:183-187)."""
        return _coherency(4, beam_i, beam_j, flux, self.device)
