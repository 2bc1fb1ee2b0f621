"""GPU parity at the sizes of BASELINE.json configs[3] and [4] (C4: 1e6 sources; C5: HERA-350 eigenbeams in
fp32), the source-chunk / memory knobs, the lattice path's entry buffers, run-to-run determinism and the
sharded multi-rank path with the GPU engine.  Every call goes through the C ABI; checks are against the CPU
oracle's exact sums on baseline subsets, the device-side brute-force sum, and size-independent properties
(linearity, chunking invariance, basis == per-antenna beams)."""

import os
import subprocess
import sys

import numpy as np
import pytest

import fftvis_amd
from fftvis_amd import synth
from fftvis_amd._lib import FftvisHipError
from fftvis_amd.gpu import gpu_nufft2d
from fftvis_amd.gpu.nufft import gpu_nudft_direct
from oracle import fftvis_oracle as orc
from tests.helpers import oracle_simulate, rel_l2

pytestmark = pytest.mark.gpu
TOL = 5 * 6e-8
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sim_c4_million_sources(gpu, monkeypatch):
    """configs[3] at its full catalog and baseline set: HERA-350, 1e6 sources, polarized table beam, all
    61 075 baselines, 2 channels (band edges) x 1 time, fp64 eps 6e-8.  A random 64-baseline subset against
    the device-side brute-force sum (O(M N), independent of spread / FFT / gather) fed with the oracle's
    strengths, a 16-baseline subset against the CPU oracle end to end, and linearity in the flux."""
    cfg = synth.make_config("C4", nfreq=2, ntimes=1)
    assert len(cfg["ra"]) == 1_000_000 and len(cfg["baselines"]) == 61075
    v = fftvis_amd.simulate_vis(**cfg)
    assert v.shape == (2, 1, 2, 2, 61075) and np.isfinite(v).all()
    rng = np.random.default_rng(11)
    sub = sorted(rng.choice(61075, 64, replace=False))
    # the CPU oracle, exact sums, on the first 16 of them
    s16 = sub[:16]
    exact = oracle_simulate(dict(cfg, baselines=[cfg["baselines"][i] for i in s16]))
    assert rel_l2(v[..., s16], exact) < TOL
    # the oracle's per-source strengths summed by brute force on the GPU for all 64
    monkeypatch.setattr(orc, "nudft_type3", lambda coords, c, targets, **kw: gpu_nudft_direct(coords, c, targets))
    brute = oracle_simulate(dict(cfg, baselines=[cfg["baselines"][i] for i in sub]))
    monkeypatch.undo()
    assert rel_l2(v[..., sub], brute) < TOL
    assert rel_l2(brute[..., :16], exact) < 1e-10  # the two checkers agree with each other
    # linearity at full size: V(a - 2 b) = V(a) - 2 V(b)
    _, _, fl2 = synth.catalog(1_000_000, cfg["freqs"], 9)
    vb = fftvis_amd.simulate_vis(**dict(cfg, fluxes=fl2))
    vab = fftvis_amd.simulate_vis(**dict(cfg, fluxes=cfg["fluxes"] - 2.0 * fl2))
    assert rel_l2(vab, v - 2.0 * vb) < 1e-9
    # the reference's source-chunk loop at this size: 4 chunks of 250 000 sources accumulate to the same block
    v4 = fftvis_amd.simulate_vis(**dict(cfg, min_chunks=4))
    assert rel_l2(v4, v) < 1e-9


def test_sim_c5_eigenbeams_fp32(gpu, monkeypatch):
    """configs[4]'s shape: HERA-350, 1e5 sources, K = 4 tabulated basis beams with per-antenna coefficients,
    precision 1, eps 1e-4, all 61 075 baselines, 2 channels x 1 time.  (a) a random subset of baselines
    against the oracle's eigenbeam path (cpu_simulate.py:303-470 restated) at the fp32 tolerance the
    reference's own eigenbeam test uses in spirit (tests/test_beam_basis.py:344-396); (b) on the 15
    baselines among 6 antennas (3 of them outriggers, so the grid stays HERA-350's): the basis run equals
    the run that gives every antenna its own beam sum_k c[a, k] B_k."""
    cfg = synth.make_config("C5", nfreq=2, ntimes=1)
    assert cfg["precision"] == 1 and cfg["eps"] == 1e-4 and len(cfg["beam"]) == 4
    v = fftvis_amd.simulate_vis(**cfg)
    assert v.dtype == np.complex64 and v.shape == (2, 1, 2, 2, 61075) and np.isfinite(v).all()
    rng = np.random.default_rng(5)
    sub = sorted(rng.choice(61075, 24, replace=False))
    exact = oracle_simulate(dict(cfg, baselines=[cfg["baselines"][i] for i in sub]))
    assert rel_l2(v[..., sub], exact) < 2e-3
    # the diagonal (k, k) terms ride the Hermitian packing (two transforms instead of four): same answer without it
    monkeypatch.setenv("FFTVIS_HIP_NO_HERMITIAN", "1")
    plain = fftvis_amd.simulate_vis(**cfg)
    monkeypatch.delenv("FFTVIS_HIP_NO_HERMITIAN")
    assert 0 < rel_l2(v, plain) < 2e-3
    # (b) basis == per-antenna beams
    antnums = list(cfg["ants"])
    chosen = [0, 57, 211, 325, 337, 349]  # core antennas and outriggers
    bl = [(a, b) for i, a in enumerate(chosen) for b in chosen[i + 1:]]
    coefs = cfg["beam_coefs"]
    freqs = cfg["freqs"]
    per_ant = []
    for a in chosen:
        tab = sum(coefs[antnums.index(a), k, :, None, None, None, None] * cfg["beam"][k].data for k in range(4))
        per_ant.append(fftvis_amd.TabulatedBeam(tab, freqs))
    beam_idx = np.zeros(len(antnums), dtype=int)
    for i, a in enumerate(chosen):
        beam_idx[antnums.index(a)] = i
    base = dict(cfg, baselines=bl)
    vb = fftvis_amd.simulate_vis(**base)
    pa = {k: base[k] for k in base if k != "beam_coefs"}
    vp = fftvis_amd.simulate_vis(**dict(pa, beam=per_ant, beam_idx=beam_idx))
    assert vb.shape == vp.shape == (2, 1, 2, 2, 15)
    assert rel_l2(vb, vp) < 2e-3
    # the same identity in fp64 at a tight tolerance pins it far below fp32 rounding
    vb64 = fftvis_amd.simulate_vis(**dict(base, precision=2, eps=1e-9))
    vp64 = fftvis_amd.simulate_vis(**dict(pa, beam=per_ant, beam_idx=beam_idx, precision=2, eps=1e-9))
    assert rel_l2(vb64, vp64) < 1e-7 and rel_l2(vb, vb64) < 2e-3


def test_sim_source_chunks_and_memory_knobs(gpu):
    """nchunks (wrapper: min_chunks / max_memory) cuts the source axis like the reference's chunk loop
    (cpu_simulate.py:939, += at :1024,1069): the result does not depend on it.  C1 in full, the C2 geometry
    (fused-gather path, gang launches), a polarized two-beam case through the stand-alone gather, the
    lattice path, the eigenbeam path; source_buffer too small raises like matvis."""
    c1 = synth.make_config("C1")
    ref = fftvis_amd.simulate_vis(**c1)
    for n in (2, 3, 100, 1000):  # more chunks than sources is clipped (wrapper.py: min(..., nsrc))
        assert rel_l2(fftvis_amd.simulate_vis(**dict(c1, min_chunks=n)), ref) < 1e-12, n
    assert rel_l2(ref, oracle_simulate(c1)) < TOL
    c2 = synth.make_config("C2", nsrc=3000, nfreq=16, ntimes=5)
    r2 = fftvis_amd.simulate_vis(**c2)
    assert rel_l2(fftvis_amd.simulate_vis(**dict(c2, min_chunks=3)), r2) < 1e-12
    # a device-memory budget too small for the whole catalog's per-time scratch forces chunks by itself
    from fftvis_amd.core.utils import get_desired_chunks

    n_auto, _ = get_desired_chunks(6_000_000, 1, [c2["beam"]], 1, 1, 37, 3000, 2, nfreq=16)
    assert n_auto > 1
    assert rel_l2(fftvis_amd.simulate_vis(**dict(c2, max_memory=6_000_000)), r2) < 1e-12
    # polarized, two table beams, flipped baselines, polarized sky (k_interp accumulate path on a big grid)
    freqs = c1["freqs"]
    tab = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, nza=46, naz=90), freqs)
    tab2 = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, diameter=12.0, nza=46, naz=90), freqs)
    _, _, fl4 = synth.catalog(100, freqs, 0, polarized_sky=True)
    pol = dict(c1, polarized=True, beam=[tab, tab2], beam_idx=np.array([0, 1, 0, 1, 1, 0, 1]), fluxes=fl4,
               baselines=c1["baselines"] + [(3, 0), (6, 1), (2, 2)])
    rp = fftvis_amd.simulate_vis(**pol)
    assert rel_l2(fftvis_amd.simulate_vis(**dict(pol, min_chunks=4)), rp) < 1e-12
    big = synth.make_config("C3", nsrc=5000, nfreq=2, ntimes=2)
    big["baselines"] = big["baselines"][::97]
    rb = fftvis_amd.simulate_vis(**big)
    assert rel_l2(fftvis_amd.simulate_vis(**dict(big, min_chunks=3)), rb) < 1e-11
    # lattice (type-1) path and eigenbeam path
    lat = {k: v for k, v in c1.items() if k != "force_use_type3"}
    rl = fftvis_amd.simulate_vis(**lat)
    assert rel_l2(fftvis_amd.simulate_vis(**dict(lat, min_chunks=3)), rl) < 1e-12
    rng = np.random.default_rng(2)
    coefs = 0.1 * (rng.normal(size=(7, 2, 8)) + 1j * rng.normal(size=(7, 2, 8)))
    coefs[:, 0] += 1.0
    rtab = [fftvis_amd.TabulatedBeam(b.data.real.astype(complex), freqs) for b in (tab, tab2)]
    bas = dict(c1, polarized=True, beam=rtab, beam_coefs=coefs)
    rbas = fftvis_amd.simulate_vis(**bas)
    assert rel_l2(fftvis_amd.simulate_vis(**dict(bas, min_chunks=3)), rbas) < 1e-12
    # source_buffer: roughly half of an isotropic catalog is up; a buffer of 10 % must raise, 0.9 must not
    with pytest.raises(FftvisHipError, match="increase source_buffer"):
        fftvis_amd.simulate_vis(**dict(c2, source_buffer=0.1))
    assert rel_l2(fftvis_amd.simulate_vis(**dict(c2, source_buffer=0.9)), r2) < 1e-12
    assert rel_l2(fftvis_amd.simulate_vis(**c2), r2) < 1e-12  # the handle recovered after the error
    with pytest.raises(ValueError, match="source_buffer"):
        fftvis_amd.simulate_vis(**dict(c2, source_buffer=1.5))
    with pytest.raises(ValueError, match="use_feed"):
        fftvis_amd.simulate_vis(**dict(c2, use_feed="z"))
    with pytest.raises(ValueError, match="needs matvis / astropy"):
        fftvis_amd.simulate_vis(**dict(c2, coord_method="CoordinateRotationERFA"))
    with pytest.raises(ValueError, match="needs matvis / astropy|unknown coord_method"):
        fftvis_amd.simulate_vis(**dict(c2, coord_method="Nope"))


def test_sim_time_blocks_and_streamed_coord_mgr(gpu, monkeypatch):
    """The engine walks the time axis in blocks when the output would not fit (here: forced), and a
    coordinate manager's vectors are streamed block by block: same visibilities either way."""
    from fftvis_amd.gpu import gpu_simulate

    cfg = synth.make_config("C2", nsrc=1200, nfreq=6, ntimes=7)
    ref = fftvis_amd.simulate_vis(**cfg)

    class Mgr:
        def __init__(self):
            self.o = orc.SimpleCoordinateRotation(None, cfg["times"], cfg["telescope_loc"], cfg["ra"], cfg["dec"])
            self.rotated = []

        def setup(self):
            pass

        def rotate(self, ti):
            self.rotated.append(ti)
            self.o.rotate(ti)
            self.all_coords_topo = self.o._topo

    monkeypatch.setattr(gpu_simulate, "_time_block", lambda *a, **k: 3)
    assert rel_l2(fftvis_amd.simulate_vis(**cfg), ref) < 1e-12
    m = Mgr()
    got = fftvis_amd.simulate_vis(**dict(cfg, coord_method="CoordinateRotationERFA"), coord_mgr=m)
    assert m.rotated == list(range(7)) and rel_l2(got, ref) < 1e-12


def test_reference_default_call_runs_end_to_end(gpu, monkeypatch):
    """the call the reference makes -- ``simulate_vis(..., backend="gpu")`` with the default
    ``coord_method="CoordinateRotationERFA"`` and no manager -- runs: the engine builds matvis' manager itself
    (stubbed here: its vectors are the oracle's sidereal ones), calls ``setup()`` once, ``rotate`` once per time
    in order, and the visibilities equal the run that applies the same rotation on the device."""
    from tests.helpers import install_reference_dependency_stubs

    made, _ = install_reference_dependency_stubs(monkeypatch)
    cfg = synth.make_config("C2", nsrc=900, nfreq=5, ntimes=4)
    ref = fftvis_amd.simulate_vis(**cfg)  # coord_method="SiderealRotation" (synthetic configs name it)
    kw = {k: v for k, v in cfg.items() if k != "coord_method"}
    got = fftvis_amd.simulate_vis(backend="gpu", min_chunks=2, **kw)
    (m,) = made
    assert m.setup_calls == 1 and m.rotated == list(range(4)) and m.kw["chunk_size"] == 450
    assert rel_l2(got, ref) < 1e-12


def test_type1_entry_buffers_hold_a_sky_that_is_all_up(gpu):
    """on small lattices (n2 = 64) every (source, frequency) pair needs ~1.5 entries at
    the default fp64 tolerance (periodic images of 14-16 cell footprints), and a catalog that is entirely
    above the horizon used to overflow entry buffers sized at 1.3 per pair.  2.4e4 sources within 50
    degrees of the zenith, HERA-37 lattice, default eps: type 1 equals type 3."""
    cfg = synth.make_config("C2", nsrc=24_000, nfreq=4, ntimes=1)
    from fftvis_amd.core.coords import gmst_rad

    rng = np.random.default_rng(4)
    lst = gmst_rad(cfg["times"][0]) + synth.HERA_LON
    th = np.deg2rad(50.0) * np.sqrt(rng.uniform(0, 1, 24_000))
    ph = rng.uniform(0, 2 * np.pi, 24_000)
    cfg["dec"] = np.clip(synth.HERA_LAT + th * np.sin(ph), -np.pi / 2, np.pi / 2)
    cfg["ra"] = lst + th * np.cos(ph) / np.cos(synth.HERA_LAT)
    cfg.update(eps=None)
    cfg.pop("force_use_type3")
    t1 = fftvis_amd.simulate_vis(**cfg)
    t3 = fftvis_amd.simulate_vis(**cfg, force_use_type3=True)
    assert np.isfinite(t1).all() and rel_l2(t1, t3) < 1e-11
    sub = cfg["baselines"][::40]
    assert rel_l2(fftvis_amd.simulate_vis(**dict(cfg, baselines=sub)), oracle_simulate(dict(cfg, baselines=sub, force_use_type3=False))) < 1e-11


def test_bad_coordinates_fail_loudly(gpu):
    """NaN source coordinates are refused by the stand-alone transform (finufft refuses them too) and
    non-unit coord_mgr vectors fail the run instead of producing finite wrong numbers."""
    rng = np.random.default_rng(0)
    x, y = rng.uniform(-3, 3, (2, 500))
    c = rng.normal(size=500) + 0j
    s, t = rng.uniform(-50, 50, (2, 40))
    gpu_nufft2d(x, y, c, s, t, 1e-6)
    xb = x.copy()
    xb[17] = np.nan
    with pytest.raises(FftvisHipError, match="NaN"):
        gpu_nufft2d(xb, y, c, s, t, 1e-6)
    assert np.isfinite(gpu_nufft2d(x, y, c, s, t, 1e-6)).all()  # the workspace survived
    cfg = synth.make_config("C1")

    class Mgr:  # vectors three times too long: outside the unit sphere's box
        def __init__(self):
            self.o = orc.SimpleCoordinateRotation(None, cfg["times"], cfg["telescope_loc"], cfg["ra"], cfg["dec"])

        def setup(self):
            pass

        def rotate(self, ti):
            self.o.rotate(ti)
            self.all_coords_topo = 3.0 * self.o._topo

    with pytest.raises(FftvisHipError, match="outside"):
        fftvis_amd.simulate_vis(**cfg, coord_mgr=Mgr())
    assert rel_l2(fftvis_amd.simulate_vis(**cfg), oracle_simulate(cfg)) < TOL


def test_runs_are_bitwise_reproducible_without_the_fused_gather(gpu):
    """The spread is a gather (every cell written once, no atomics), the FFT passes and the stand-alone
    gather have a fixed summation order: two runs of the same transform / simulation are BIT-identical.
    Only the fused gather of small 2-D grids adds partial sums with fp64 atomics (compiled with
    -munsafe-fp-atomics) and is reproducible to rounding, not bitwise."""
    rng = np.random.default_rng(1)
    x, y = rng.uniform(-3, 3, (2, 20_000))
    c = rng.normal(size=(8, 20_000)) + 1j * rng.normal(size=(8, 20_000))
    s, t = rng.uniform(-300, 300, (2, 3000))
    a = gpu_nufft2d(x, y, c, s, t, 6e-8)
    b = gpu_nufft2d(x, y, c, s, t, 6e-8)
    assert np.array_equal(a, b)
    big = synth.make_config("C3", nsrc=30_000, nfreq=3, ntimes=2)  # 8192^2-class grid: stand-alone gather
    big["baselines"] = big["baselines"][::13]
    assert np.array_equal(fftvis_amd.simulate_vis(**big), fftvis_amd.simulate_vis(**big))
    c2 = synth.make_config("C2", nsrc=3000, nfreq=16, ntimes=4)  # fused gather: atomics
    u, v = fftvis_amd.simulate_vis(**c2), fftvis_amd.simulate_vis(**c2)
    assert rel_l2(u, v) < 1e-13


def test_sharded_run_two_ranks_on_the_gpu(gpu, tmp_path):
    """Two ranks (gloo rendezvous, both on device 0 -- a one-GPU box cannot host two RCCL ranks) run
    parallel.simulate_vis_sharded through the GPU engine: catalog broadcast into device memory, one
    cost-balanced (time x freq) block per rank via GPUSimulationEngine(time_idx, freq_idx), blocks
    assembled on rank 0 (reference cpu_simulate.py:800-847) == the single-process GPU result."""
    out = tmp_path / "sharded.npz"
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29731",
           os.path.join(ROOT, "tests", "sharded_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    z = np.load(out)
    cfg = synth.make_config("C2", nsrc=2500, nfreq=12, ntimes=6)
    cfg["polarized"] = True
    freqs = cfg["freqs"]
    cfg["beam"] = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, nza=46, naz=90), freqs)
    single = fftvis_amd.simulate_vis(**cfg)
    assert z["vis"].shape == single.shape == (12, 6, 2, 2, 666)
    assert rel_l2(z["vis"], single) < 1e-12      # blocks delivered straight into one shared-memory result
    assert rel_l2(z["vis_p2p"], single) < 1e-12  # blocks sent to rank 0 point to point
    assert [tuple(b) for b in z["blocks"]] == [(0, 3, 0, 12), (3, 6, 0, 12)]


def test_bench_gpus_n_starts_its_own_ranks(gpu):
    """`python bench.py --gpus 2` started as ONE process spawns its two ranks itself (a child
    torchrun; gloo rendezvous and both ranks on device 0 here, because a one-GPU box cannot host two RCCL ranks) and
    prints a line with n_gpus = 2 and one per-rank time per rank; a rank count that differs from --gpus aborts."""
    import json

    env = dict(os.environ, PYTHONPATH=ROOT, FFTVIS_BENCH_BACKEND="gloo", FFTVIS_BENCH_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "5",
           "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and len(res["config"]["per_rank_ms_per_step"]) == 2
    assert res["config"]["finite_output"] and res["value"] > 0 and res["scaling"] == "strong"
    bad = subprocess.run(cmd[:2] + ["--gpus", "2", "--workload", "C2", "--no-cpu-baseline"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True,
                         timeout=300)
    assert bad.returncode != 0 and "rank(s) joined" in bad.stderr + bad.stdout


def _band_block(name, f0, f1, ntimes, seed=0):
    """Channels [f0, f1) of the FULL band of a BASELINE configuration (its channel spacing, hence its frequency
    groups), the full catalog and every baseline, ``ntimes`` time steps."""
    _, ns, nf, _, _, _ = synth.CONFIGS[name]
    cfg = synth.make_config(name, nsrc=1000, nfreq=2, ntimes=ntimes)  # skeleton: array, times, baselines
    freqs = np.linspace(100e6, 200e6, nf)[f0:f1]
    cfg["ra"], cfg["dec"], cfg["fluxes"] = synth.catalog(ns, freqs, seed)
    cfg["freqs"] = freqs
    cfg["beam"] = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs), freqs)
    return cfg


@pytest.mark.parametrize("name,f0,f1,ntimes,nsub", [("C3", 96, 128, 2, 16), ("C3", 0, 32, 2, 16), ("C4", 240, 256, 1, 8)])
def test_launches_the_bench_times_are_parity_checked(gpu, monkeypatch, name, f0, f1, ntimes, nsub):
    """what the driver's bench line times -- C3 / C4 geometry, the full catalog, all
    61 075 baselines on the device, frequency groups of 8 / 16 / 24 packed transforms under the default grid
    budget (k_spread2d<.., 8|16> at C3's density, k_spread2d_mm<8|16> at C4's 10^6 sources), gang launches as
    the engine picks them -- checked: a subset of baselines against the CPU oracle (exact sums), ALL baselines
    against the four-transform run (FFTVIS_HIP_NO_HERMITIAN=1: different launches, same answer)."""
    from fftvis_amd.gpu import gpu_simulate

    cfg = _band_block(name, f0, f1, ntimes)
    nch = f1 - f0
    gpu_simulate.release_handles()
    monkeypatch.setenv("FFTVIS_HIP_HANDLE_CACHE_BYTES", str(2**40))  # keep the handle: its counters are read below
    v = fftvis_amd.simulate_vis(**cfg)
    assert v.shape == (nch, ntimes, 2, 2, 61075) and np.isfinite(v).all()
    (h,) = gpu_simulate._IDLE_HANDLES.values()
    st = h.stats()
    # launches are counted per (time, frequency group): the groups hold 8, 16 or 24 packed transforms
    per_launch = 2.0 * nch * ntimes / st["spread_launches"]
    assert 8 <= per_launch <= 24 and st["spread_launches"] >= 2 * ntimes, st
    sub = sorted(np.random.default_rng(17).choice(61075, nsub, replace=False))
    exact = oracle_simulate(dict(cfg, baselines=[cfg["baselines"][i] for i in sub]))
    assert rel_l2(v[..., sub], exact) < TOL
    monkeypatch.setenv("FFTVIS_HIP_NO_HERMITIAN", "1")
    plain = fftvis_amd.simulate_vis(**cfg)
    monkeypatch.delenv("FFTVIS_HIP_NO_HERMITIAN")
    assert 0 < rel_l2(v, plain) < TOL
    for a in range(2):
        for b in range(2):
            assert rel_l2(v[:, :, a, b], plain[:, :, a, b]) < 4 * TOL, (a, b)
    gpu_simulate.release_handles()


def test_hermitian_packing_matches_four_transforms(gpu, monkeypatch):
    """Single-beam polarized runs on large grids pack their Hermitian strengths into two transforms per
    frequency (c_00 + i c_11, c_01) and rebuild the four products from the baseline and its mirror image
    (k_interp<.., HERM>).  Against the plain four-transform run (FFTVIS_HIP_NO_HERMITIAN=1) and the oracle:
    unpolarized sky, polarized sky (complex c_01), a table beam with complex leakage, flipped baselines and
    autos, a non-coplanar array (3-D transform), source chunks, fp32; two different real-valued beams (the
    cross pair's strengths are all real: c_01 + i c_10 share a transform too), a real and a complex beam."""
    cfg = synth.make_config("C3", nsrc=20_000, nfreq=3, ntimes=2)
    bl = cfg["baselines"][::23] + [(5, 5), (340, 2), (349, 17)]   # autos, and pairs given "backwards"
    cfg["baselines"] = bl
    _, _, fl4 = synth.catalog(20_000, cfg["freqs"], 3, polarized_sky=True)
    tilted = {k: v + np.array([0.0, 0.0, 0.02 * v[0] + 0.3 * np.sin(0.01 * v[1])]) for k, v in cfg["ants"].items()}
    freqs = cfg["freqs"]
    real_a = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, 14.0).real.astype(complex), freqs)
    real_b = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, 12.0).real.astype(complex), freqs)
    bidx = np.arange(len(cfg["ants"])) % 2
    two_real = dict(cfg, beam=[real_a, real_b], beam_idx=bidx)   # pairs (0,0), (1,1): Hermitian; (0,1): all-real packing
    cases = {"unpolarized sky": cfg, "polarized sky": dict(cfg, fluxes=fl4), "two real beams": two_real,
             "real and complex beam": dict(two_real, beam=[real_a, cfg["beam"]]),   # (0,1) keeps four transforms
             "non-coplanar": dict(cfg, ants=tilted, baselines=bl[::9]),
             "chunks": dict(cfg, min_chunks=3), "fp32": dict(cfg, precision=1, eps=1e-4)}
    sub = list(range(0, len(bl), 40)) + [len(bl) - 3, len(bl) - 2, len(bl) - 1]
    for name, c in cases.items():
        monkeypatch.delenv("FFTVIS_HIP_NO_HERMITIAN", raising=False)
        packed = fftvis_amd.simulate_vis(**c)
        monkeypatch.setenv("FFTVIS_HIP_NO_HERMITIAN", "1")
        plain = fftvis_amd.simulate_vis(**c)
        tol = 2e-3 if name == "fp32" else TOL
        d = rel_l2(packed, plain)
        assert 0 < d < tol, (name, d)   # two different computations (not the same path twice), same answer
        # every one of the four products separately, not just the block as a whole
        for a in range(2):
            for b in range(2):
                assert rel_l2(packed[:, :, a, b], plain[:, :, a, b]) < 4 * tol, (name, a, b)
        if name in ("unpolarized sky", "polarized sky", "two real beams"):
            cs = dict(c, baselines=[c["baselines"][i] for i in sub])
            assert rel_l2(packed[..., sub], oracle_simulate(cs)) < TOL, name
    monkeypatch.delenv("FFTVIS_HIP_NO_HERMITIAN", raising=False)


def test_redundant_baselines_are_gathered_once(gpu, monkeypatch):
    """A regular array repeats most of its baseline vectors (HERA-350: 61 075 baselines, < 8 000 distinct vectors).
    The gather evaluates each distinct (beam pair, sign-adjusted vector) once and writes every member's slot, with
    the member's own conjugation / feed transposition.  Against the run that gathers every baseline by itself
    (FFTVIS_HIP_NO_TARGET_DEDUP=1) -- packed and four-transform launches, two beams with baselines given
    "backwards" in both symmetry modes, a non-coplanar array, source chunks, fp32 -- and against the oracle on a
    subset that holds members of the largest runs; exact duplicates in the caller's list come back bit-equal."""
    from fftvis_amd.gpu import gpu_simulate

    cfg = synth.make_config("C3", nsrc=20_000, nfreq=3, ntimes=2)
    nant = len(cfg["ants"])
    bl = list(cfg["baselines"])
    rng = np.random.default_rng(5)
    back = rng.choice(len(bl), 4000, replace=False)
    for i in back:  # given "backwards": the engine flips them, conjugates (and, exact mode, transposes) on the way out
        bl[i] = (bl[i][1], bl[i][0])
    bl += [bl[7], bl[7], (3, 3), (9, 9)]  # exact duplicates and autos (all autos share the vector 0)
    cfg["baselines"] = bl
    freqs = cfg["freqs"]
    other = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, 12.0), freqs)
    two = dict(cfg, beam=[cfg["beam"], other], beam_idx=np.arange(nant) % 2)
    tilted = {k: v + np.array([0.0, 0.0, 0.02 * v[0] + 0.3 * np.sin(0.01 * v[1])]) for k, v in cfg["ants"].items()}
    cases = {"packed": (cfg, {}), "four transforms": (cfg, {"FFTVIS_HIP_NO_HERMITIAN": "1"}),
             "two beams": (two, {}), "two beams, exact symmetries": (dict(two, reference_compat=False), {}),
             "non-coplanar": (dict(cfg, ants=tilted, baselines=bl[::7]), {}),
             "chunks": (dict(cfg, min_chunks=3), {}), "fp32": (dict(cfg, precision=1, eps=1e-4), {})}
    for name, (c, env) in cases.items():
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        gpu_simulate.release_handles()
        monkeypatch.setenv("FFTVIS_HIP_HANDLE_CACHE_BYTES", str(2**40))
        monkeypatch.delenv("FFTVIS_HIP_NO_TARGET_DEDUP", raising=False)
        once = fftvis_amd.simulate_vis(**c)
        (h,) = gpu_simulate._IDLE_HANDLES.values()
        items_once = h.stats()["interp_items"]
        h.reset_stats()
        monkeypatch.setenv("FFTVIS_HIP_NO_TARGET_DEDUP", "1")
        each = fftvis_amd.simulate_vis(**c)
        items_each = h.stats()["interp_items"]
        monkeypatch.delenv("FFTVIS_HIP_NO_TARGET_DEDUP")
        for k_ in env:
            monkeypatch.delenv(k_)
        # coplanar fp64 lists: > 3x fewer footprints; the tilted array has few repeats and may keep the plain list, and
        # fp32 baselines carry their own rounding (6e-8 of the longest), far above the merging tolerance: fewer runs
        lim = {"non-coplanar": 1.0001, "fp32": 0.9}.get(name, 0.3)
        assert items_once < lim * items_each, (name, items_once, items_each)
        tol = 1e-6 if name == "fp32" else 1e-3 * 6e-8  # the members' vectors differ by rounding: far below eps
        assert rel_l2(once, each) < tol, (name, rel_l2(once, each))
        if name in ("packed", "two beams"):
            n = len(c["baselines"])
            assert np.array_equal(once[..., n - 4], once[..., n - 3]) and np.array_equal(once[..., n - 4], once[..., 7])
            sub = sorted(set(rng.choice(n - 4, 12, replace=False)) | {int(back[0]), int(back[1]), n - 4, n - 2, n - 1})
            cs = dict(c, baselines=[c["baselines"][i] for i in sub])
            assert rel_l2(once[..., sub], oracle_simulate(cs)) < TOL, name
    gpu_simulate.release_handles()


def test_column_plan_stores_only_the_columns_targets_read(gpu, monkeypatch):
    """The baselines of a regular array sit on a lattice, so their gather footprints meet only a fraction of the
    transform's columns in the first dimension (HERA-350: about a third).  The engine plans those columns per
    (frequency group, beam pair) on the host, the x-pass stores only them (compacted), the y-pass transforms only
    them and the gather finds them through the plan's table.  Same arithmetic on fewer columns: against the run
    without a plan (FFTVIS_HIP_NO_COLUMN_PLAN=1) the block must agree to rounding -- packed and four-transform
    launches, two beams with flipped baselines (exact symmetries: transposed slots), eigenbeams with the mirror
    gather, source chunks, gang launches over 3 time steps, fp32 -- with fewer FFT cells moved; a random layout
    (no repeated vectors) keeps every column and runs unplanned."""
    from fftvis_amd.gpu import gpu_simulate

    cfg = synth.make_config("C3", nsrc=20_000, nfreq=3, ntimes=3)
    nant = len(cfg["ants"])
    bl = list(cfg["baselines"])
    rng = np.random.default_rng(6)
    for i in rng.choice(len(bl), 3000, replace=False):
        bl[i] = (bl[i][1], bl[i][0])
    cfg["baselines"] = bl
    freqs = cfg["freqs"]
    other = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, 12.0), freqs)
    two = dict(cfg, beam=[cfg["beam"], other], beam_idx=np.arange(nant) % 2, reference_compat=False)
    c5 = synth.make_config("C5", nsrc=20_000, nfreq=2, ntimes=1)
    c5x = dict(c5, beam=[fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(c5["freqs"], 14.0 * (1 + 0.05 * i), nza=91, naz=180), c5["freqs"])
                         for i in range(synth.C5_NBASIS)], reference_compat=False)  # complex basis beams: mirror gather
    scattered = {k: np.append(rng.uniform(-430, 430, 2), 0.0) for k in cfg["ants"]}  # no lattice: every column is read
    cases = {"packed": (cfg, {}, True), "four transforms": (cfg, {"FFTVIS_HIP_NO_HERMITIAN": "1"}, True),
             "two beams, exact symmetries": (two, {}, True), "eigenbeams": (c5, {}, True),
             "eigenbeams, mirror gather": (c5x, {}, True),
             "chunks": (dict(cfg, min_chunks=3), {}, True), "fp32": (dict(cfg, precision=1, eps=1e-4), {}, True),
             "scattered antennas": (dict(cfg, ants=scattered), {}, False)}
    for name, (c, env, planned) in cases.items():
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        gpu_simulate.release_handles()
        monkeypatch.setenv("FFTVIS_HIP_HANDLE_CACHE_BYTES", str(2**40))
        monkeypatch.setenv("FFTVIS_HIP_GRID_SLACK", "0")  # one geometry for both runs (a plan's geometry takes no grid slack)
        monkeypatch.delenv("FFTVIS_HIP_NO_COLUMN_PLAN", raising=False)
        some = fftvis_amd.simulate_vis(**c)
        (h,) = gpu_simulate._IDLE_HANDLES.values()
        cells_some = h.stats()["fft_cells"]
        h.reset_stats()
        monkeypatch.setenv("FFTVIS_HIP_NO_COLUMN_PLAN", "1")
        every = fftvis_amd.simulate_vis(**c)
        cells_every = h.stats()["fft_cells"]
        monkeypatch.delenv("FFTVIS_HIP_NO_COLUMN_PLAN")
        monkeypatch.delenv("FFTVIS_HIP_GRID_SLACK")
        for k_ in env:
            monkeypatch.delenv(k_)
        assert np.isfinite(some).all()
        if planned:
            assert cells_some < 0.8 * cells_every, (name, cells_some, cells_every)
        else:
            assert cells_some == cells_every, (name, cells_some, cells_every)
        d = rel_l2(some, every)
        assert d < (1e-6 if c.get("precision") == 1 else 1e-13), (name, d)
    gpu_simulate.release_handles()


def test_source_disc_prunes_spread_blocks_and_row_loads(gpu, monkeypatch):
    """Sources are projections of unit vectors onto the array plane: inside the disc |x| <= 2 pi whatever the box.
    The spread then launches only the blocks the disc can reach and the x-pass reads each row over the disc's
    chord only.  Nothing but never-touched zeros is left out: the block must equal the run without the disc
    (FFTVIS_HIP_NO_DISC=1) to rounding -- HERA-350 geometry (folded rows, paired residues), a band whose x-pass
    runs Q = 4096 rows, C2 (small grids, fused gather), a planar array on a slope (rotated plane: the disc is
    off-centre in the box), fp32."""
    from fftvis_amd.gpu import gpu_simulate

    cfg = synth.make_config("C3", nsrc=20_000, nfreq=3, ntimes=2)
    cfg["baselines"] = cfg["baselines"][::5]
    mid = dict(cfg, freqs=np.linspace(140e6, 142e6, 3))
    mid["beam"] = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(mid["freqs"]), mid["freqs"])
    slope = {k: v + np.array([0.0, 0.0, 0.05 * v[0] - 0.02 * v[1]]) for k, v in cfg["ants"].items()}
    c2 = synth.make_config("C2", nfreq=8, ntimes=2)
    cases = {"C3": cfg, "Q = 4096 rows": mid, "C2": c2, "slope": dict(cfg, ants=slope), "fp32": dict(cfg, precision=1, eps=1e-4)}
    for name, c in cases.items():
        gpu_simulate.release_handles()
        monkeypatch.delenv("FFTVIS_HIP_NO_DISC", raising=False)
        pruned = fftvis_amd.simulate_vis(**c)
        gpu_simulate.release_handles()
        monkeypatch.setenv("FFTVIS_HIP_NO_DISC", "1")
        full = fftvis_amd.simulate_vis(**c)
        monkeypatch.delenv("FFTVIS_HIP_NO_DISC")
        assert np.isfinite(pruned).all()
        d = rel_l2(pruned, full)
        assert d < (1e-6 if c.get("precision") == 1 else 1e-13), (name, d)
    gpu_simulate.release_handles()


def _random_regular_array_config(rng):
    """Random regular array on a LARGE grid (the regime of the column plan): hex / rectangular lattices with random spacing,
    holes, rotation, a tilt, outriggers or a few displaced antennas; random baseline subsets with reversed pairs and
    autos, 1-3 channels, 1-2 times, one or two beams, polarized or not, eps in [1e-9, 1e-5], source chunks."""
    kind = rng.choice(["hex", "rect", "hex+out", "jitter"])
    sp, nside = rng.uniform(8, 30), int(rng.integers(6, 14))
    pts = []
    for a in range(-nside, nside + 1):
        for b in range(-nside, nside + 1):
            if kind != "rect":
                if abs(a + b) <= nside:
                    pts.append((sp * (a + 0.5 * b), sp * np.sqrt(3) / 2 * b, 0.0))
            else:
                pts.append((sp * a, sp * 0.8 * b, 0.0))
    pts = np.array(pts)
    pts = pts[rng.uniform(size=len(pts)) < rng.uniform(0.3, 0.9)][: int(rng.integers(40, 160))]
    if kind == "hex+out":
        pts = np.vstack([pts, rng.uniform(-3, 3, (6, 3)) * sp * nside * np.array([1, 1, 0])])
    if kind == "jitter":
        pts = pts + np.append(rng.uniform(-0.3, 0.3, 2), 0) * (rng.uniform(size=(len(pts), 1)) < 0.1)
    rot = rng.uniform(0, np.pi)
    R = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1]])
    tilt = rng.uniform(-0.05, 0.05, 2) if rng.uniform() < 0.3 else np.zeros(2)
    ants = {i: R @ q + np.array([0, 0, tilt[0] * q[0] + tilt[1] * q[1]]) for i, q in enumerate(pts)}
    nant = len(ants)
    allb = [(i, j) for i in range(nant) for j in range(i, nant)]
    sel = rng.choice(len(allb), size=min(len(allb), int(rng.integers(200, 4000))), replace=False)
    bl = [allb[k] if rng.uniform() < 0.7 else allb[k][::-1] for k in sel]
    nfreq, ntimes = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    freqs = np.sort(rng.uniform(120e6, 260e6) * (1 + rng.uniform(0, 0.05, nfreq)))
    times = np.linspace(2459845.0, 2459845.0 + rng.uniform(0.001, 0.1), ntimes)
    pol = bool(rng.uniform() < 0.6)
    ra, dec, flux = synth.catalog(int(rng.integers(500, 4000)), freqs, int(rng.integers(1e6)), polarized_sky=pol and rng.uniform() < 0.4)
    nbeam = 1 if rng.uniform() < 0.6 else 2
    beams = [fftvis_amd.AiryBeam(float(rng.uniform(6, 16))) if rng.uniform() < 0.4 else
             fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, float(rng.uniform(8, 15)), nza=46, naz=90), freqs)
             for _ in range(nbeam)]
    return dict(ants=ants, fluxes=flux, ra=ra, dec=dec, freqs=freqs, times=times, beam=beams if nbeam > 1 else beams[0],
                beam_idx=rng.integers(0, nbeam, nant) if nbeam > 1 else None, telescope_loc=(synth.HERA_LAT, synth.HERA_LON),
                baselines=bl, polarized=pol, precision=2, eps=float(10 ** rng.uniform(-9, -5)), force_use_type3=True,
                coord_method="SiderealRotation", reference_compat=bool(rng.uniform() < 0.7), min_chunks=int(rng.integers(1, 3)))


def test_prunings_fuzz_on_large_regular_arrays(gpu, monkeypatch):
    """Seeded fuzz of the prunings of DESIGN section 2 together: 24 random regular arrays on grids of 10^6 ... 10^8 cells,
    the pruned run against the run with column plan, redundant-baseline gather and source disc switched off, and
    against the oracle on a 24-baseline subset -- all to the run's tolerance.  (640 more configurations of the same
    generator, `scratch/fuzz_plan.py`, were run clean while writing it: worst error 1.0 eps.)"""
    from fftvis_amd.gpu import gpu_simulate

    rng = np.random.default_rng(7)
    off = ("FFTVIS_HIP_NO_COLUMN_PLAN", "FFTVIS_HIP_NO_TARGET_DEDUP", "FFTVIS_HIP_NO_DISC")
    planned = 0
    monkeypatch.setenv("FFTVIS_HIP_HANDLE_CACHE_BYTES", str(2**40))
    for it in range(24):
        cfg = _random_regular_array_config(rng)
        gpu_simulate.release_handles()
        for k in off:
            monkeypatch.delenv(k, raising=False)
        v = fftvis_amd.simulate_vis(**cfg)
        (h,) = gpu_simulate._IDLE_HANDLES.values()
        cells = h.stats()["fft_cells"]
        gpu_simulate.release_handles()
        for k in off:
            monkeypatch.setenv(k, "1")
        w = fftvis_amd.simulate_vis(**cfg)
        (h,) = gpu_simulate._IDLE_HANDLES.values()
        planned += cells < 0.8 * h.stats()["fft_cells"]
        for k in off:
            monkeypatch.delenv(k)
        sub = sorted(rng.choice(len(cfg["baselines"]), size=min(len(cfg["baselines"]), 24), replace=False))
        exact = oracle_simulate(dict(cfg, baselines=[cfg["baselines"][i] for i in sub]))
        assert np.isfinite(v).all()
        assert rel_l2(v, w) < 10 * cfg["eps"], (it, rel_l2(v, w), cfg["eps"])
        assert rel_l2(v[..., sub], exact) < 10 * cfg["eps"] + 1e-12, (it, rel_l2(v[..., sub], exact), cfg["eps"])
    assert planned >= 5, planned  # the generator does reach the planned regime
    gpu_simulate.release_handles()


def test_nufft2d_planes_of_4_gib_take_the_transpose_path(gpu):
    """A fine grid whose planes pass 4 GiB per transform (36864 x 32768 cells here; the column pass addresses a
    plane with 32-bit byte offsets): the engine falls back to the tile transpose + row pass for such planes and
    the result still meets the tolerance.  Exact sum at a handful of targets."""
    from oracle import nudft

    rng = np.random.default_rng(5)
    M, N = 60, 40
    x, y = rng.uniform(-3, 3, (2, M))
    c = rng.normal(size=(1, M)) + 1j * rng.normal(size=(1, M))
    s = rng.uniform(-4400, 4400, N)
    t = rng.uniform(-4280, 4280, N)
    got = gpu_nufft2d(x, y, c, s, t, 1e-9)
    assert rel_l2(got, nudft.nudft_type3([x, y], c, [s, t])) < 5e-9


def test_fft_layout_switches_leave_the_result_bit_identical(gpu, tmp_path):
    """The column-blocked layout of the x-pass output moves data, not arithmetic: switched off
    (FFTVIS_HIP_NO_BLOCKED_B) the transform returns the same bits.  (The switch is read once per process, hence
    the worker.)"""
    from tests.nufft_worker import problem

    ref = gpu_nufft2d(*problem(), 1e-9)
    out = tmp_path / "plain.npy"
    env = dict(os.environ, PYTHONPATH=ROOT, FFTVIS_HIP_NO_BLOCKED_B="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nufft_worker.py"), str(out)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert np.array_equal(np.load(out), ref)


def test_paired_residue_jobs_leave_the_result_unchanged(gpu, tmp_path):
    """The folded row FFT can compute two residues per job from one pass over a row's inputs (k_rowfft_st<.., PAIR>;
    default: row mode).  Off (FFTVIS_HIP_PAIR=0), row mode only (1) and row + column mode (2) agree to rounding on a
    grid whose two passes are both folded (5 and 4 residues: an odd count leaves a job with one residue), and meet the
    tolerance against the exact sum.  (The switch is read once per process, hence the worker.)"""
    from oracle import nudft
    from tests.nufft_worker import problem_folded

    x, y, c, s, t = problem_folded()
    exact = nudft.nudft_type3([x, y], c, [s, t])
    res = {}
    for mode in ("0", "1", "2"):
        out = tmp_path / f"pair{mode}.npy"
        env = dict(os.environ, PYTHONPATH=ROOT, FFTVIS_HIP_PAIR=mode)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nufft_worker.py"), str(out), "folded"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        res[mode] = np.load(out)
        assert rel_l2(res[mode], exact) < 5e-9, mode
    assert rel_l2(res["1"], res["0"]) < 1e-13 and rel_l2(res["2"], res["0"]) < 1e-13


@pytest.mark.gpu
def test_bench_takes_the_rccl_code_path_with_one_rank(gpu):
    """What the driver's N > 1 runs do that no one-GPU box can: RCCL.  `FFTVIS_BENCH_FORCE_DIST=1` sends ONE rank down
    the N > 1 code path with the real backend ("nccl" = RCCL): process group on the device, catalog broadcast into device
    memory, barriers, MAX-reduce and object collectives of the timing, and the sharded host-to-host calls
    (`e2e_sharded`: per-rank flux columns, shared-memory result) -- everything but a second rank."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, FFTVIS_BENCH_FORCE_DIST="1", FFTVIS_BENCH_NO_PMC_CHECK="1")
    env.pop("FFTVIS_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--nsrc", "3000", "--nfreq", "8",
                        "--ntimes", "2", "--steps", "1", "--warmup", "1", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["finite_output"]
    e = line["e2e_sharded"]
    assert "error" not in e and e["finite_output"] and e["second_call_s"] > 0, e
    assert "cpu_baseline" in line
