"""Regenerates tests/golden/*.npz.

Two kinds of vectors:
 * ``coherency_*.npz``: inputs + expected outputs of the four per-source 2x2 coherency products,
   where the EXPECTED values come from ``np.einsum`` with the index strings the reference's own
   tests use as their oracle (tests/test_cpu_beams.py:102,563,578,606,870,953) on the inputs those
   tests specify (arange / literal matrices / default_rng seeds).  No reference code is run or
   copied; einsum is the independent check.
 * ``sim_c1_*.npz``: BASELINE.json configs[0] ("C1": HERA-7, 100 sources, 8 freqs, 2 times,
   unpolarized Airy) inputs and the visibilities produced by oracle/fftvis_oracle.py (exact direct
   sum).  These pin the oracle against regressions and give the GPU tests a fixture that does
   not depend on the oracle code being importable.

Run:  python tests/golden/make_golden.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def cplx(rng, shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def coherency_vectors():
    cases = {}
    # (A^H A) I  -- einsum "bas,s,bcs->acs" (tests/test_cpu_beams.py:99-102, 552-563, 566-578, 337-361)
    beam = np.arange(12).reshape((2, 2, 3)).astype(complex)
    flux = np.arange(3).astype(float)
    cases["beam_arange"] = (0, beam, beam, flux, np.einsum("bas,s,bcs->acs", beam.conj(), flux, beam))
    bc = np.array([[[1 + 2j, 3 + 4j], [5 + 6j, 7 + 8j]],
                   [[9 + 10j, 11 + 12j], [13 + 14j, 15 + 16j]]]).transpose(1, 2, 0)
    fl = np.array([2.0, 3.0])
    cases["beam_literal"] = (0, bc, bc, fl, np.einsum("bas,s,bcs->acs", bc.conj(), fl, bc))
    bz = np.zeros((2, 2, 3), dtype=complex)
    bz[0, 0, 0], bz[1, 1, 1], bz[0, 1, 2] = 1, 2, 3
    fz = np.array([1.5, 2.5, 3.5])
    cases["beam_zeros"] = (0, bz, bz, fz, np.einsum("bas,s,bcs->acs", bz.conj(), fz, bz))
    b1 = np.ones((2, 2, 1), dtype=complex)
    cases["beam_single"] = (0, b1, b1, np.array([2.0]), np.ones((2, 2, 1)) * 4.0)
    # A^H C A -- einsum 'kin,kmn,mjn->ijn' (tests/test_cpu_beams.py:592-606)
    C = np.array([[[2 + 1j, 4 + 3j], [6 + 5j, 8 + 7j]],
                  [[10 + 9j, 12 + 11j], [14 + 13j, 16 + 15j]]]).transpose(1, 2, 0)
    cases["polsky_literal"] = (1, bc, bc, C, np.einsum("kin,kmn,mjn->ijn", bc.conj(), C, bc))
    # Ai^H Aj I -- einsum "bas,s,bps->aps", seeds 0..4 (tests/test_cpu_beams.py:879-934)
    for seed, n in [(0, 5), (1, 8), (2, 4), (3, 1), (4, 6)]:
        rng = np.random.default_rng(seed)
        if seed == 0:
            bi = cplx(rng, (2, 2, n))
            bj = bi.copy()
            f = rng.standard_normal(n)
        else:
            bi, bj = cplx(rng, (2, 2, n)), cplx(rng, (2, 2, n))
            f = np.zeros(n) if seed == 2 else rng.standard_normal(n)
            if seed == 4:
                f = np.abs(f)
        cases[f"pair_seed{seed}"] = (2, bi, bj, f, np.einsum("bas,s,bps->aps", bi.conj(), f, bj))
    # Ai^H C Aj -- einsum "bas,bks,kps->aps", seeds 10..15 (tests/test_cpu_beams.py:967-1023)
    for seed, n in [(11, 7), (12, 5), (13, 6), (14, 1), (15, 5)]:
        rng = np.random.default_rng(seed)
        if seed == 12:
            bi = cplx(rng, (2, 2, n))
            bj = bi.copy()
            Cq = cplx(rng, (2, 2, n))
        elif seed == 15:
            bi, bj = cplx(rng, (2, 2, n)), cplx(rng, (2, 2, n))
            f = rng.standard_normal(n)
            Cq = np.zeros((2, 2, n), dtype=complex)
            Cq[0, 0] = Cq[1, 1] = f / 2
        else:
            bi, bj = cplx(rng, (2, 2, n)), cplx(rng, (2, 2, n))
            Cq = np.zeros((2, 2, n), dtype=complex) if seed == 13 else cplx(rng, (2, 2, n))
        cases[f"polpair_seed{seed}"] = (3, bi, bj, Cq, np.einsum("bas,bks,kps->aps", bi.conj(), Cq, bj))
    rng = np.random.default_rng(10)
    ident = np.zeros((2, 2, 4), dtype=complex)
    ident[0, 0] = ident[1, 1] = 1
    Cq = cplx(rng, (2, 2, 4))
    cases["polpair_identity"] = (3, ident, ident, Cq, Cq.copy())
    flat = {}
    for k, (variant, bi, bj, fl, exp) in cases.items():
        flat[k + "__variant"] = np.array(variant)
        flat[k + "__beam_i"], flat[k + "__beam_j"] = bi, bj
        flat[k + "__flux"], flat[k + "__expected"] = fl, exp
    np.savez(os.path.join(HERE, "coherency_cases.npz"), **flat)


def sim_vectors():
    from fftvis_amd import synth
    from tests.helpers import oracle_simulate

    cfg = synth.make_config("C1")
    out = oracle_simulate(cfg)
    cfgp = dict(cfg, polarized=True)
    outp = oracle_simulate(cfgp)
    np.savez(
        os.path.join(HERE, "sim_c1.npz"),
        antpos=np.array(list(cfg["ants"].values())), fluxes=cfg["fluxes"], ra=cfg["ra"],
        dec=cfg["dec"], freqs=cfg["freqs"], times=cfg["times"],
        telescope_loc=np.array(cfg["telescope_loc"]), baselines=np.array(cfg["baselines"]),
        airy_diameter=np.array(14.0), vis_unpolarized=out, vis_polarized=outp,
    )


if __name__ == "__main__":
    coherency_vectors()
    sim_vectors()
    print("golden vectors written to", HERE)
