"""Abstract engine interface (mirror of the reference's src/fftvis/core/simulate.py:16-221)."""

from __future__ import annotations

from abc import ABC, abstractmethod

# Default NUFFT accuracy by precision (reference core/simulate.py:16-19).
default_accuracy_dict = {1: 6e-8, 2: 1e-13}


class SimulationEngine(ABC):
    """Base class for visibility simulation engines (reference core/simulate.py:22)."""

    @abstractmethod
    def simulate(self, ants, freqs, fluxes, beam_list, ra, dec, times, telescope_loc, **kw):
        """Return visibilities (nfreqs, ntimes, nbls) or (nfreqs, ntimes, 2, 2, nbls)."""

    @abstractmethod
    def _evaluate_vis_chunk(self, time_idx, freq_idx, **kw):
        """Return the (nt_here, nbls, nfeeds, nfeeds, nf_here) block of one task."""
