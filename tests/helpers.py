"""Shared helpers for the parity tests: oracle twins of the product's beam containers."""

import numpy as np

import fftvis_amd
from fftvis_amd.core.beams import spline_order
from oracle import fftvis_oracle as orc


def rel_l2(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / (nb if nb else 1.0)


def oracle_beam(beam, polarized, freqs, order=1):
    if isinstance(beam, fftvis_amd.AiryBeam):
        return orc.AiryBeam(beam.diameter, "efield" if polarized else "power")
    tb = beam if polarized else beam.power_from_efield()
    return orc.TabulatedBeam(tb.data, freqs, tb.za_max, "efield" if polarized else "power", order)


def oracle_simulate(cfg):
    """Run the oracle on simulate_vis-style keyword arguments."""
    beams = cfg["beam"] if isinstance(cfg["beam"], list) else [cfg["beam"]]
    order = spline_order(cfg.get("beam_spline_opts"))
    ob = [oracle_beam(b, cfg["polarized"], cfg["freqs"], order) for b in beams]
    return orc.simulate(
        cfg["ants"], cfg["freqs"], cfg["fluxes"], ob, cfg["ra"], cfg["dec"], cfg["times"],
        cfg["telescope_loc"], baselines=cfg.get("baselines"), beam_idx=cfg.get("beam_idx"),
        polarized=cfg["polarized"], beam_coefs=cfg.get("beam_coefs"),
        force_use_type3=cfg.get("force_use_type3", True),
    )
