"""Eigenbeam preprocessing (host side) -- mirrors the reference's tests/test_beam_basis.py:
shape/rank properties (:82-153), reconstruction (:176-255), argument validation (:262-300).
The end-to-end check (basis simulation == per-antenna-beam simulation, :344-396) is the GPU test
at the bottom."""

import numpy as np
import pytest

import fftvis_amd
from fftvis_amd import AiryBeam, TabulatedBeam, compute_beam_basis, compute_beam_basis_per_freq
from fftvis_amd import synth

FREQ = 150e6


@pytest.fixture(scope="module")
def beam_a():
    return AiryBeam(14.0)


@pytest.fixture(scope="module")
def beam_b():
    return AiryBeam(12.0)


def _kw():
    return dict(n_axis1=73, n_axis2=37)


def test_single_beam_returns_one_eigenbeam(beam_a):
    eb, coefs = compute_beam_basis([beam_a], FREQ, True, **_kw())
    assert len(eb) == 1 and coefs.shape == (1, 1)
    assert eb[0].is_efield and eb[0].data.shape == (1, 2, 2, 37, 72)


def test_coefs_shape(beam_a, beam_b):
    eb, coefs = compute_beam_basis([beam_a, beam_b], FREQ, True, **_kw())
    assert coefs.shape == (2, len(eb))


def test_custom_axes(beam_a):
    az = np.linspace(0, 2 * np.pi, 41)
    za = np.linspace(0, np.pi / 2, 19)
    eb, _ = compute_beam_basis([beam_a], FREQ, True, axis1_array=az, axis2_array=za)
    assert eb[0].data.shape[-2:] == (19, 40)   # the repeated 2 pi node is dropped (periodic table)
    assert np.isclose(eb[0].za_max, np.pi / 2)


def test_identical_beams_yield_one_mode(beam_a):
    eb, coefs = compute_beam_basis([beam_a] * 3, FREQ, True, threshold=1e-10, **_kw())
    assert len(eb) == 1 and coefs.shape == (3, 1)
    np.testing.assert_allclose(coefs[0], coefs[1], rtol=1e-12)


def test_threshold_monotone(beam_a, beam_b):
    _, tight = compute_beam_basis([beam_a, beam_b], FREQ, True, threshold=1e-12, **_kw())
    _, loose = compute_beam_basis([beam_a, beam_b], FREQ, True, threshold=0.5, **_kw())
    assert tight.shape[1] >= 2
    assert loose.shape[1] <= tight.shape[1]


@pytest.mark.parametrize("polarized", [True, False])
def test_full_rank_reconstruction(beam_a, beam_b, polarized):
    beams = [beam_a, beam_b, AiryBeam(9.0)]
    eb, coefs = compute_beam_basis(beams, FREQ, polarized, **_kw())
    rec = np.tensordot(coefs, np.stack([e.data[0] for e in eb]), axes=(1, 0))
    for i, b in enumerate(beams):
        one, c1 = compute_beam_basis([b], FREQ, polarized, **_kw())
        np.testing.assert_allclose(rec[i], c1[0, 0] * one[0].data[0], atol=1e-12)


def test_table_beams_resampled_and_chromatic():
    freqs = np.array([140e6, 160e6])
    tabs = [TabulatedBeam(synth.synthetic_efield_table(freqs, d, nza=46, naz=90), freqs) for d in (14.0, 13.0)]
    eb, coefs = compute_beam_basis(tabs, 150e6, True)
    assert eb[0].data.shape == (1, 2, 2, 46, 90)
    mid = 0.5 * (tabs[0].data[0] + tabs[0].data[1])     # linear in frequency
    rec = np.tensordot(coefs[0], np.stack([e.data[0] for e in eb]), axes=(0, 0))
    np.testing.assert_allclose(rec, mid, atol=1e-12)
    # coarser common grid: bilinear resampling hits the nodes it shares with the table
    eb2, c2 = compute_beam_basis(tabs, 150e6, True, axis1_array=np.linspace(0, 2 * np.pi, 46),
                                 axis2_array=np.linspace(0, np.pi, 16))
    rec2 = np.tensordot(c2[0], np.stack([e.data[0] for e in eb2]), axes=(0, 0))
    np.testing.assert_allclose(rec2, mid[:, :, ::3, ::2], atol=1e-12)


def test_per_freq_layout(beam_a, beam_b):
    freqs = np.linspace(120e6, 180e6, 3)
    eb, coefs = compute_beam_basis_per_freq([beam_a, beam_b, beam_a], freqs, True, 2, **_kw())
    assert len(eb) == 2 and eb[0].data.shape == (3, 2, 2, 37, 72) and coefs.shape == (3, 2, 3)
    for fi, f in enumerate(freqs):
        one, c1 = compute_beam_basis([beam_b], float(f), True, **_kw())
        rec = sum(coefs[1, k, fi] * eb[k].data[fi] for k in range(2))
        np.testing.assert_allclose(rec, c1[0, 0] * one[0].data[0], atol=1e-12)


def test_errors(beam_a):
    with pytest.raises(ValueError, match="beam_list must contain at least one beam"):
        compute_beam_basis([], FREQ, True)
    for thr in (0.0, 1.5):
        with pytest.raises(ValueError, match="threshold must be in the interval"):
            compute_beam_basis([beam_a], FREQ, True, threshold=thr)
    with pytest.raises(ValueError, match="scalar freq"):
        compute_beam_basis([beam_a], [1e8, 2e8], True)
    with pytest.raises(ValueError, match="must be supplied together"):
        compute_beam_basis([beam_a], FREQ, True, axis1_array=np.linspace(0, 2 * np.pi, 10))
    with pytest.raises(ValueError, match="must be supplied together"):
        compute_beam_basis([beam_a], FREQ, True, axis2_array=np.linspace(0, np.pi, 10))
    power = TabulatedBeam(np.ones((1, 10, 20)))
    with pytest.raises(ValueError, match="requires efield beams"):
        compute_beam_basis([power], FREQ, True)


@pytest.mark.gpu
def test_basis_simulation_matches_per_antenna_beams(gpu):
    """Reference tests/test_beam_basis.py:370-396: a simulation through the SVD basis reproduces
    the simulation that evaluates every antenna's own beam."""
    cfg = synth.make_config("C1", nsrc=60, nfreq=3, ntimes=2)
    freqs = cfg["freqs"]
    nant = len(cfg["ants"])
    diam = 14.0 * (1 + 0.07 * np.linspace(-1, 1, nant))
    # real-valued Jones tables: the (l, k) = (k, l)^T shortcut of the basis path is exact only
    # for those (reference cpu_simulate.py:464-468)
    beams = [TabulatedBeam(synth.synthetic_efield_table(freqs, d, nza=91, naz=180).real.astype(complex), freqs)
             for d in diam]
    eb, coefs = compute_beam_basis_per_freq(beams, freqs, True, nant)   # full rank: exact
    base = dict(cfg, polarized=True, eps=1e-10)
    base.pop("beam")
    ref = fftvis_amd.simulate_vis(beam=beams, beam_idx=np.arange(nant), **base)
    got = fftvis_amd.simulate_vis(beam=eb, beam_coefs=coefs, **base)
    assert got.shape == ref.shape == (len(freqs), 2, 2, 2, len(cfg["baselines"]))
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert err < 1e-8, err           # two NUFFT evaluations at eps = 1e-10
    # truncated basis: error follows the discarded singular values (reference notebook
    # beam_decomposition.ipynb:633-638: K=3 1.5e-5 ... K=6 3.6e-10 for its Gaussian family)
    eb5, c5 = compute_beam_basis_per_freq(beams, freqs, True, 5)
    got5 = fftvis_amd.simulate_vis(beam=eb5, beam_coefs=c5, **base)
    err5 = np.linalg.norm(got5 - ref) / np.linalg.norm(ref)
    assert 1e-9 < err5 < 5e-5, err5
