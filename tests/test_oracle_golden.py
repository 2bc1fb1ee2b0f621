"""Pins the CPU oracle (oracle/) against the reference's own known-answer checks.

Each test restates a check from the reference's test-suite (file:line cited) or a committed
fixture under tests/golden (tests/golden/make_golden.py).  CPU only.
"""

import os

import numpy as np
import pytest

from oracle import cpu_nufft, nudft
from oracle import fftvis_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cases():
    z = np.load(os.path.join(GOLD, "coherency_cases.npz"))
    names = sorted({k.split("__")[0] for k in z.files})
    return z, names


def _run_oracle_kernel(variant, bi, bj, fl):
    if variant == 0:
        b = bi.copy()
        orc.get_apparent_flux_polarized_beam(b, fl)
        return b
    if variant == 1:
        b = bi.copy()
        orc.get_apparent_flux_polarized(b, fl)
        return b
    out = np.zeros_like(bi)
    if variant == 2:
        orc.get_apparent_flux_polarized_beam_pair(bi, bj, fl, out)
    else:
        orc.get_apparent_flux_polarized_pair(bi, bj, fl, out)
    return out


@pytest.mark.parametrize("name", _cases()[1])
def test_coherency_kernels_match_reference_einsum_vectors(name):
    """reference tests/test_cpu_beams.py:90-109,337-361,541-607,861-1023 (rtol 1e-12)."""
    z, _ = _cases()
    got = _run_oracle_kernel(int(z[name + "__variant"]), z[name + "__beam_i"], z[name + "__beam_j"],
                             z[name + "__flux"])
    np.testing.assert_allclose(got, z[name + "__expected"], rtol=1e-12, atol=1e-14)


def test_coherency_empty_input():
    """reference tests/test_cpu_beams.py:337-347: zero sources is a no-op."""
    b = np.zeros((2, 2, 0), dtype=complex)
    orc.get_apparent_flux_polarized_beam(b, np.zeros(0))
    assert b.shape == (2, 2, 0)


def test_pair_result_not_hermitian_for_distinct_beams():
    """reference tests/test_cpu_beams.py:927-935."""
    rng = np.random.default_rng(4)
    bi = rng.standard_normal((2, 2, 6)) + 1j * rng.standard_normal((2, 2, 6))
    bj = rng.standard_normal((2, 2, 6)) + 1j * rng.standard_normal((2, 2, 6))
    out = np.zeros_like(bi)
    orc.get_apparent_flux_polarized_beam_pair(bi, bj, np.abs(rng.standard_normal(6)), out)
    assert not np.allclose(out[0, 1], np.conj(out[1, 0]))


class TestPrepareBeamEvaluation:
    """Truth tables of reference tests/test_cpu_beams.py:715-854."""

    def test_none_beam_idx(self):
        p, i, f = orc.prepare_beam_evaluation([0, 1, 2], [(0, 1), (1, 2), (0, 2)], None)
        assert p == [(0, 0)]
        np.testing.assert_array_equal(i[(0, 0)], np.arange(3))
        assert f[(0, 0)] == [False, False, False]

    def test_single_type(self):
        p, i, f = orc.prepare_beam_evaluation([0, 1, 2], [(0, 1), (1, 2), (0, 2)], [0, 0, 0])
        assert p == [(0, 0)] and list(i[(0, 0)]) == [0, 1, 2] and f[(0, 0)] == [False] * 3

    def test_two_types_and_flips(self):
        p, i, f = orc.prepare_beam_evaluation([0, 1], [(0, 1)], [0, 1])
        assert set(p) == {(0, 0), (0, 1), (1, 1)}
        assert i[(0, 1)] == [0] and f[(0, 1)] == [False]
        _, i, f = orc.prepare_beam_evaluation([0, 1], [(1, 0)], [0, 1])
        assert i[(0, 1)] == [0] and f[(0, 1)] == [True]
        _, i, f = orc.prepare_beam_evaluation([0, 1], [(0, 1), (1, 0)], [0, 1])
        assert i[(0, 1)] == [0, 1] and f[(0, 1)] == [False, True]

    def test_many_baselines_one_pair(self):
        _, i, f = orc.prepare_beam_evaluation(
            [0, 1, 2, 3], [(0, 2), (0, 3), (1, 2), (1, 3)], [0, 0, 1, 1])
        assert sorted(i[(0, 1)]) == [0, 1, 2, 3] and f[(0, 1)] == [False] * 4

    def test_empty_baselines(self):
        p, i, f = orc.prepare_beam_evaluation([0, 1], [], [0, 1])
        assert all(i[bp] == [] and f[bp] == [] for bp in p)

    def test_three_types_and_noncontiguous(self):
        p, _, _ = orc.prepare_beam_evaluation([0, 1, 2], [(0, 1), (0, 2), (1, 2)], [0, 1, 2])
        assert len(p) == 6
        p, i, f = orc.prepare_beam_evaluation([0, 1, 2], [(0, 1), (0, 2), (1, 2)], [0, 2, 2])
        assert len(p) == 3 and 0 in i[(0, 2)] and 2 in i[(2, 2)]
        assert f[(0, 2)][i[(0, 2)].index(0)] is False


def test_core_utils_known_answers():
    """reference tests/test_core_utils.py:26-170."""
    n, fc, tc, nf, nt = orc.get_task_chunks(3, 30, 1)
    assert len(fc) == len(tc) == n and nf == 10 and nt == 1
    assert {i for c in fc for i in range(*c.indices(30))} == set(range(30))
    n, fc, tc, nf, nt = orc.get_task_chunks(10, 5, 1)
    assert n == 1 and len(fc) == 1 and nf == 5 and nt == 1
    ants = {0: np.zeros(3), 1: np.array([10.0, 0, 0]), 2: np.array([0, 10.0, 0]),
            3: np.array([-10.0, 0, 0]), 4: np.array([0, -10.0, 0])}
    r = orc.get_pos_reds(ants, include_autos=False)
    assert len(r) == 6 and sum(map(len, r)) == 10
    r = orc.get_pos_reds(ants, include_autos=True)
    assert len(r) == 7 and sum(map(len, r)) == 15
    rng = np.random.default_rng(0)
    av = rng.normal(size=(20, 3)) * [50, 50, 0.5]
    R = orc.get_plane_to_xy_rotation_matrix(av)
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
    assert np.isclose(np.linalg.det(R), 1.0)
    assert np.array_equal(orc.get_plane_to_xy_rotation_matrix(av * [1, 1, 0]), np.eye(3))
    b = np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]).T.copy()
    rot = np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])
    orc.inplace_rot(rot, b)
    np.testing.assert_allclose(b, rot @ np.eye(3))


def test_source_catalog_rules():
    """reference cpu/utils.py:26-80 / tests/test_wrapper.py:123-141."""
    f = np.arange(6.0).reshape(3, 2)
    c, pol = orc.prepare_source_catalog(f, False)
    assert not pol and np.array_equal(c, 0.5 * f)
    s = np.random.default_rng(0).normal(size=(3, 2, 4))
    c, pol = orc.prepare_source_catalog(s, True)
    assert pol and c.shape == (3, 2, 2, 2)
    I, Q, U, V = (s[..., k] for k in range(4))
    np.testing.assert_allclose(c[..., 0, 0], 0.5 * (I + Q))
    np.testing.assert_allclose(c[..., 0, 1], 0.5 * (U + 1j * V))
    np.testing.assert_allclose(c[..., 1, 0], 0.5 * (U - 1j * V))
    np.testing.assert_allclose(c[..., 1, 1], 0.5 * (I - Q))
    with pytest.raises(ValueError, match="requires sky_model to be 2D"):
        orc.prepare_source_catalog(s, False)
    with pytest.raises(ValueError, match="polarized_beam=True requires"):
        orc.prepare_source_catalog(s[..., :3], True)


def test_c_nudft_equals_numpy_nudft():
    rng = np.random.default_rng(5)
    x, y, z = rng.uniform(-6, 6, (3, 257))
    c = rng.normal(size=(3, 257)) + 1j * rng.normal(size=(3, 257))
    s, t, u = rng.uniform(-40, 40, (3, 91))
    for d in (2, 3):
        a = nudft.nudft_type3([x, y, z][:d], c, [s, t, u][:d])
        b = orc.nudft_type3([x, y, z][:d], c, [s, t, u][:d])
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-10)
    a = nudft.nudft_type3([x, y], c[0], [s, t], isign=-1)
    np.testing.assert_allclose(a, np.conj(nudft.nudft_type3([x, y], np.conj(c[0]), [s, t])), atol=1e-10)


@pytest.mark.parametrize("eps", [1e-3, 6e-8, 1e-12])
@pytest.mark.parametrize("dim", [2, 3])
def test_cpu_nufft_port_meets_eps(eps, dim):
    """The CPU port used as bench baseline agrees with the exact sum to ~eps."""
    rng = np.random.default_rng(6)
    M, N = 1500, 300
    X = list(rng.uniform(-2 * np.pi, 2 * np.pi, (dim, M)))
    S = list(rng.uniform(-30, 30, (dim, N)))
    if dim == 3:
        S[2] = S[2] / 30
    c = rng.normal(size=(2, M)) + 1j * rng.normal(size=(2, M))
    ex = nudft.nudft_type3(X, c, S)
    got = cpu_nufft.nufft_type3(X, c, S, eps=eps)
    assert np.linalg.norm(got - ex) / np.linalg.norm(ex) < 10 * eps


def test_oracle_sim_fixture_c1():
    """The committed C1 fixture is what the oracle produces today (regression pin)."""
    from tests.helpers import oracle_simulate
    from fftvis_amd import synth

    z = np.load(os.path.join(GOLD, "sim_c1.npz"))
    cfg = synth.make_config("C1")
    np.testing.assert_allclose(np.array(list(cfg["ants"].values())), z["antpos"])
    np.testing.assert_allclose(oracle_simulate(cfg), z["vis_unpolarized"], rtol=1e-10, atol=1e-12)
    assert z["vis_unpolarized"].shape == (8, 2, 21)       # (nfreqs, ntimes, nbls)
    assert z["vis_polarized"].shape == (8, 2, 2, 2, 21)   # tests/test_cpu_simulate.py:184-189


def test_oracle_sim_structure():
    """Structural pins of the reference: identical beams == one beam (tests/test_cpu_simulate.py
    :273-382), source chunking is a no-op, Airy-polarized xx equals unpolarized power sim."""
    from fftvis_amd import synth

    cfg = synth.make_config("C1", nsrc=40, nfreq=2, ntimes=2)
    args = (cfg["ants"], cfg["freqs"], cfg["fluxes"])
    kw = dict(ra=cfg["ra"], dec=cfg["dec"], times=cfg["times"], telescope_loc=cfg["telescope_loc"],
              baselines=cfg["baselines"])
    one = orc.simulate(*args, [orc.AiryBeam(14.0, "power")], **kw)
    two = orc.simulate(*args, [orc.AiryBeam(14.0, "power"), orc.AiryBeam(14.0, "power")],
                       beam_idx=np.array([0, 1, 0, 1, 0, 1, 0]), **kw)
    np.testing.assert_allclose(one, two, rtol=1e-12, atol=1e-14)
    chunked = orc.simulate(*args, [orc.AiryBeam(14.0, "power")], nchunks=3, **kw)
    np.testing.assert_allclose(one, chunked, rtol=1e-12, atol=1e-14)
    pol = orc.simulate(*args, [orc.AiryBeam(14.0, "efield")], polarized=True, **kw)
    # all four Jones entries equal e  =>  (A^H A)[0,0] = 2 e^2
    np.testing.assert_allclose(pol[:, :, 0, 0, :], 2 * one, rtol=1e-12, atol=1e-14)
    diff = orc.simulate(*args, [orc.AiryBeam(14.0, "power"), orc.AiryBeam(7.0, "power")],
                        beam_idx=np.array([0, 1, 0, 1, 0, 1, 0]), **kw)
    assert not np.allclose(one, diff)


def test_oracle_order3_table_is_scipy_map_coordinates():
    """The oracle's order-3 table interpolation (what pyuvdata's az_za_map_coordinates hands to
    scipy, cpu/beams.py:69-74) against scipy.ndimage.map_coordinates itself: the periodic az axis is
    emulated by padding a full period on both sides (the prefilter's boundary influence decays as
    0.268^k), za uses scipy's "mirror"; and the spline reproduces the table at its nodes."""
    from scipy.ndimage import map_coordinates

    rng = np.random.default_rng(0)
    nza, naz, za_max = 23, 36, 1.6
    tab = rng.normal(size=(1, nza, naz))
    b = orc.TabulatedBeam(tab, [1e8], za_max=za_max, beam_type="power", order=3)
    az = rng.uniform(-7, 7, 500)
    za = rng.uniform(0, za_max, 500)
    za[:3] = [0, za_max, 0.8]
    got = b.compute_response(az, za, [1e8])[0, 0, 0].real
    pad = np.concatenate([tab[0]] * 3, axis=1)
    fa = np.mod(az, 2 * np.pi) / (2 * np.pi / naz) + naz
    ref = map_coordinates(pad, [za / (za_max / (nza - 1)), fa], order=3, mode="mirror")
    np.testing.assert_allclose(got, ref, atol=1e-12)
    zi, ai = np.meshgrid(np.arange(nza), np.arange(naz), indexing="ij")
    nodes = b.compute_response((ai * 2 * np.pi / naz).ravel(), (zi * za_max / (nza - 1)).ravel(), [1e8])
    np.testing.assert_allclose(nodes[0, 0, 0].real, tab[0].ravel(), atol=1e-12)
    # complex Jones tables: real and imaginary parts are filtered separately
    jt = rng.normal(size=(2, 2, 2, nza, naz)) + 1j * rng.normal(size=(2, 2, 2, nza, naz))
    bj = orc.TabulatedBeam(jt, [1e8, 2e8], za_max=za_max, order=3)
    r = bj.compute_response(az, za, [2e8])
    padj = np.concatenate([jt[1, 1, 0].imag] * 3, axis=1)
    np.testing.assert_allclose(r[1, 0, 0].imag, map_coordinates(padj, [za / (za_max / (nza - 1)), fa], order=3,
                                                                mode="mirror"), atol=1e-12)


def test_astrometry_restatement_against_erfa_where_it_is_installed():
    """the per-source astrometry the device applies (fv_astrom_topo) is pinned to oracle/astrometry.py,
    written from the published algorithm descriptions.  Where PyERFA is importable (never in the build pipeline, which
    is why this skips there) the restatement is held against ERFA itself: ``erfa.apco13`` fills the context,
    ``erfa.atciqz`` + ``erfa.atioq`` are the reference's per-source chain (matvis CoordinateRotationERFA), with and
    without refraction."""
    erfa = pytest.importorskip("erfa")
    from oracle import astrometry as oa

    rng = np.random.default_rng(4)
    ra = rng.uniform(0, 2 * np.pi, 2000)
    dec = np.arcsin(rng.uniform(-1, 1, 2000))
    p = np.stack([np.cos(dec) * np.cos(ra), np.cos(dec) * np.sin(ra), np.sin(dec)])
    elong, phi, hm = np.deg2rad(21.4283), np.deg2rad(-30.7215), 1050.0
    for phpa in (0.0, 880.0):
        astrom, _eo, _j = erfa.apco13(2459845.0, 0.3, 0.05, elong, phi, hm, 1e-7, 2e-7, phpa, 15.0, 0.3, 200.0)
        a = astrom if astrom.shape == () else astrom[()]
        ctx = np.concatenate([[a["pmt"]], np.ravel(a["eb"]), np.ravel(a["eh"]), [a["em"]], np.ravel(a["v"]), [a["bm1"]],
                              np.ravel(a["bpn"]), [a["along"], a["phi"], a["xpl"], a["ypl"], a["sphi"], a["cphi"],
                                                   a["diurab"], a["eral"], a["refa"], a["refb"]]]).astype(float)
        assert ctx.shape == (31,)
        ri, di = erfa.atciqz(ra, dec, astrom)
        aob, zob, _hob, _dob, _rob = erfa.atioq(ri, di, astrom)
        want = np.stack([np.sin(aob) * np.sin(zob), np.cos(aob) * np.sin(zob), np.cos(zob)])  # east, north, up
        got = oa.icrs_to_enu(p, ctx)
        up = want[2] > 0.1  # above the refraction formula's low-elevation guard
        assert np.abs(got[:, up] - want[:, up]).max() < 1e-9, phpa


@pytest.mark.parametrize("order", [0, 1, 2, 3, 4, 5])
def test_tabulated_beam_orders_against_scipy_map_coordinates(order):
    """The oracle's table interpolant (and the product's host twin of the device interpolant) for every order
    scipy.ndimage.map_coordinates takes -- what pyuvdata's az_za_map_coordinates runs for the reference
    (cpu/beams.py:69-74).  scipy has one boundary mode for all axes, the table needs two (az periodic, za mirrored):
    the table extended by its own mirror image in za is periodic in BOTH, so ``mode="grid-wrap"`` on it is exactly
    the same interpolant."""
    from scipy.ndimage import map_coordinates

    from fftvis_amd.core.beams import _interp_table

    rng = np.random.default_rng(3)
    nza, naz = 37, 48
    tab = rng.normal(size=(nza, naz)) + 1j * rng.normal(size=(nza, naz))
    ext = np.concatenate([tab, tab[-2:0:-1]], axis=0)  # period 2 (nza - 1) in za
    az, za = rng.uniform(-1, 8, 500), rng.uniform(0, np.pi, 500)
    za[:5] = [0, np.pi, 1e-9, np.pi - 1e-9, np.pi / 2]
    az[5:8] = [0.0, 2 * np.pi, 2 * np.pi / naz * 7]
    fa, fz = np.mod(az, 2 * np.pi) / (2 * np.pi / naz), za / (np.pi / (nza - 1))
    ref = (map_coordinates(ext.real, [fz, fa], order=order, mode="grid-wrap")
           + 1j * map_coordinates(ext.imag, [fz, fa], order=order, mode="grid-wrap"))
    beam = orc.TabulatedBeam(tab[None], [1e8], np.pi, "power", order)
    got = beam.compute_response(az_array=az, za_array=za, freq_array=np.array([1e8]))[0, 0, 0]
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    np.testing.assert_allclose(_interp_table(tab, np.pi, az, za, order), ref, rtol=0, atol=1e-12)
