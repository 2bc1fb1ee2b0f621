"""Round-4 GPU checks: the two new bench workloads (3-D transforms on a non-coplanar HERA-350, the generic
type-3 engine on an array without a lattice) against the CPU oracle at full catalog size, the host-output
fallback when the driver refuses to pin part of the caller's array, and fp32-rounded unit vectors at the
horizon of an fp64 run.  Everything goes through the C ABI."""

import os

import numpy as np
import pytest

import fftvis_amd
from fftvis_amd import synth
from oracle import fftvis_oracle as orc
from tests.helpers import oracle_beam, oracle_simulate, rel_l2

pytestmark = pytest.mark.gpu
TOL = 5 * 6e-8


def _top_of_band(cfg, nch):
    """The last ``nch`` channels of a configuration (its largest grids), catalog and beam table cut to match."""
    f = slice(len(cfg["freqs"]) - nch, len(cfg["freqs"]))
    out = dict(cfg, freqs=cfg["freqs"][f], fluxes=cfg["fluxes"][:, f])
    b = cfg["beam"]
    out["beam"] = fftvis_amd.TabulatedBeam(b.data[f], cfg["freqs"][f])
    return out


def test_bench_workload_c3z_three_dimensional_transform_matches_the_oracle(gpu):
    """`bench.py --workload C3z`: HERA-350 with a 3 cm height scatter -- |b_z| up to 0.17 m, far beyond the
    reference's flat_array_tol (cpu_simulate.py:655), so the run takes the 3-D transform (cpu/nufft.py:62-118):
    full catalog, all 61 075 baselines, the two top channels, one time; a random subset of baselines against the
    oracle's exact sums (its non-coplanar branch), and the height scatter must matter at the tested accuracy."""
    cfg = _top_of_band(synth.make_config("C3", ntimes=1, z_scatter=0.03), 2)
    from fftvis_amd.gpu.gpu_simulate import prepare_array

    _, bls, coplanar = prepare_array(cfg["ants"], cfg["baselines"], 1e-6, np.float64)
    assert not coplanar and np.abs(bls[2]).max() * orc.speed_of_light > 0.1
    v = fftvis_amd.simulate_vis(**cfg)
    assert v.shape == (2, 1, 2, 2, 61075) and np.isfinite(v).all()
    sub = sorted(np.random.default_rng(3).choice(61075, 16, replace=False))
    sel = dict(cfg, baselines=[cfg["baselines"][i] for i in sub])
    exact = oracle_simulate(sel)
    assert rel_l2(v[..., sub], exact) < TOL
    flat = oracle_simulate(dict(sel, ants={k: np.array([p[0], p[1], 0.0]) for k, p in cfg["ants"].items()}))
    assert rel_l2(flat, exact) > 100 * TOL  # the z term is not noise


def test_bench_workload_scattered_array_matches_the_oracle(gpu, monkeypatch):
    """`bench.py --array scattered`: 350 antennas without a lattice -- every one of the 61 075 baseline vectors is
    distinct, so the column plan keeps (nearly) every column and is not used, and no gather item serves two
    baselines.  Full catalog, all baselines, two top channels, one time, against the oracle on a subset."""
    from fftvis_amd.gpu import gpu_simulate

    cfg = _top_of_band(synth.make_config("C3", ntimes=1, array="scattered350"), 2)
    a = np.array(list(cfg["ants"].values()))
    d = (a[None, :, :2] - a[:, None, :2]).reshape(-1, 2)
    d = d[np.abs(d).sum(1) > 0]
    assert len(np.unique(np.round(d, 6), axis=0)) == len(d)  # no repeated baseline vector
    gpu_simulate.release_handles()
    monkeypatch.setenv("FFTVIS_HIP_HANDLE_CACHE_BYTES", str(2**40))
    v = fftvis_amd.simulate_vis(**cfg)
    (h,) = gpu_simulate._IDLE_HANDLES.values()
    st = h.stats()
    # packed transforms are gathered at s and -s: 2 footprints per (baseline, transform) when nothing is shared
    assert st["interp_items"] >= 0.99 * 2 * 61075 * 4, st
    gpu_simulate.release_handles()
    assert v.shape == (2, 1, 2, 2, 61075) and np.isfinite(v).all()
    sub = sorted(np.random.default_rng(5).choice(61075, 16, replace=False))
    exact = oracle_simulate(dict(cfg, baselines=[cfg["baselines"][i] for i in sub]))
    assert rel_l2(v[..., sub], exact) < TOL


def test_partial_pin_refusal_leaves_nothing_pinned_and_the_result_intact(gpu, monkeypatch):
    """the caller's array is pinned in 256-MiB pieces; a refusal after the first piece used to
    leave that piece registered under ONE pageable copy of the whole range.  Forced here
    (FFTVIS_HIP_PIN_FAIL_AFTER=1) on a 312-MB output that spans at least two pieces: the helper releases what it
    registered, the fallback copy runs over plain pageable memory, and the block equals the pinned run's bit for bit
    (large grids: no atomics anywhere)."""
    cfg = synth.make_config("C3", nsrc=2000, nfreq=8, ntimes=10)
    ref = fftvis_amd.simulate_vis(**cfg)
    assert ref.nbytes > (256 << 20)  # more than one 256-MiB piece whatever the alignment
    monkeypatch.setenv("FFTVIS_HIP_PIN_FAIL_AFTER", "1")
    got = fftvis_amd.simulate_vis(**cfg)
    monkeypatch.setenv("FFTVIS_HIP_PIN_FAIL_AFTER", "0")
    got0 = fftvis_amd.simulate_vis(**cfg)
    monkeypatch.delenv("FFTVIS_HIP_PIN_FAIL_AFTER")
    assert np.array_equal(got, ref) and np.array_equal(got0, ref)
    # the caller's memory is ordinary memory again: registering it anew must succeed piece by piece
    assert np.array_equal(fftvis_amd.simulate_vis(**cfg), ref)


def test_float32_rounded_unit_vectors_at_the_horizon_run_in_fp64(gpu):
    """a coord_mgr whose vectors were computed in float32 has norms off by 6e-8; with sources at the
    horizon their projections fall outside the unit disc by that much, which the source-disc pruning (margin 1e-9
    then) took for out-of-box sources and failed the whole precision=2 run.  The margin is 1e-6 now."""
    cfg = _top_of_band(synth.make_config("C3", nsrc=3000, ntimes=1), 1)

    class Mgr:
        def __init__(self):
            self.o = orc.SimpleCoordinateRotation(None, cfg["times"], cfg["telescope_loc"], cfg["ra"], cfg["dec"])

        def setup(self):
            pass

        def rotate(self, ti):
            self.o.rotate(ti)
            t = np.array(self.o._topo, dtype=np.float64)
            ph = np.linspace(0, 2 * np.pi, 64, endpoint=False)
            t[:, :64] = np.stack([np.cos(ph), np.sin(ph), np.full(64, 1e-9)])  # 64 sources on the horizon
            self.all_coords_topo = t.astype(np.float32).astype(np.float64)

    m = Mgr()
    m.rotate(0)
    r2 = (m.all_coords_topo[:2, :64] ** 2).sum(0)
    assert (r2 > 1 + 4e-9).any()  # beyond the old margin
    got = fftvis_amd.simulate_vis(**cfg, coord_mgr=Mgr())
    assert np.isfinite(got).all()
    sub = cfg["baselines"][::4000]
    o = Mgr()

    # the oracle with the very same vectors
    class OMgr(orc.SimpleCoordinateRotation):
        def rotate(self, ti):
            super().rotate(ti)
            o.rotate(ti)
            self._topo = o.all_coords_topo

    beams = [orc.TabulatedBeam(cfg["beam"].data, cfg["freqs"], cfg["beam"].za_max, "efield", 1)]
    coh, pol_sky = orc.prepare_source_catalog(cfg["fluxes"], True)
    exact = orc.simulate(cfg["ants"], cfg["freqs"], cfg["fluxes"], beams, cfg["ra"], cfg["dec"], cfg["times"],
                         cfg["telescope_loc"], baselines=sub, polarized=True, force_use_type3=True,
                         coord_mgr=OMgr(coh, cfg["times"], cfg["telescope_loc"], cfg["ra"], cfg["dec"]))
    idx = [cfg["baselines"].index(b) for b in sub]
    assert rel_l2(got[..., idx], exact) < TOL


def _last_handle_stats():
    from fftvis_amd.gpu import gpu_simulate

    (h,) = gpu_simulate._IDLE_HANDLES.values()
    return h.stats()


def _height_cases():
    c = synth.make_config("C2", nsrc=700, nfreq=3, ntimes=2)
    ants = synth.with_z_scatter(c["ants"], 0.05, seed=3)
    freqs = c["freqs"]
    tab = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, nza=46, naz=90), freqs)
    tab2 = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, diameter=12.0, nza=46, naz=90), freqs)
    _, _, fl4 = synth.catalog(700, freqs, 0, polarized_sky=True)
    nant = len(ants)
    bidx = np.arange(nant) % 2
    keys = list(ants)
    bls = c["baselines"][::5] + [(keys[5], keys[1]), (keys[30], keys[2]), (keys[4], keys[4])]  # flipped pairs, an auto
    rng = np.random.default_rng(12)
    coefs = rng.normal(size=(nant, 2, len(freqs))) + 1j * rng.normal(size=(nant, 2, len(freqs)))
    base = dict(c, ants=ants)
    return {
        "unpolarized_airy": base,
        "polarized_table": dict(base, polarized=True, beam=tab),
        "two_beams_flipped_polarized_sky": dict(base, polarized=True, beam=[tab, tab2], beam_idx=bidx, baselines=bls, fluxes=fl4),
        "source_chunks": dict(base, min_chunks=3),
        "eigenbeams": dict(base, polarized=True, beam=[tab, tab2], beam_coefs=coefs, baselines=bls),
        "fp32": dict(base, precision=1, eps=1e-4),
        "default_baselines": dict(base, baselines=None),
    }


@pytest.mark.parametrize("name", list(_height_cases()))
def test_height_terms_replace_the_third_grid_dimension(gpu, monkeypatch, name):
    """A non-coplanar array whose heights are small against the wavelength (HERA-37 with 5 cm of scatter) runs as K 2-D
    transforms per slice -- the expansion of exp(i z s_z) about the middle of the sources' height range, every
    baseline adding its own factor in the gather -- instead of a 3-D grid (reference: finufft.nufft3d3 whenever
    |b_z| > 1e-6 m, cpu_simulate.py:655, cpu/nufft.py:62-118).  Against the oracle's exact 3-D sums, and against the
    engine's own 3-D transform (FFTVIS_HIP_NO_WTERM=1), for every kind of run the gather serves."""
    from fftvis_amd.gpu import gpu_simulate

    cfg = _height_cases()[name]
    eps = cfg["eps"]
    gpu_simulate.release_handles()
    monkeypatch.setenv("FFTVIS_HIP_HANDLE_CACHE_BYTES", str(2**40))
    got = fftvis_amd.simulate_vis(**cfg)
    st = _last_handle_stats()
    assert 2 <= st["height_terms"] <= 16 and st["n2_3"] == 1, st  # 2-D grids, K terms
    gpu_simulate.release_handles()
    exact = oracle_simulate(cfg)
    assert got.shape == exact.shape and rel_l2(got, exact) < 5 * eps
    monkeypatch.setenv("FFTVIS_HIP_NO_WTERM", "1")
    grid3 = fftvis_amd.simulate_vis(**cfg)
    st3 = _last_handle_stats()
    assert st3["height_terms"] == 0 and st3["n2_3"] > 1, st3
    gpu_simulate.release_handles()
    assert rel_l2(got, grid3) < 6 * eps
    # the heights matter at this accuracy: the flat array is a different answer
    if name == "unpolarized_airy":
        flat = oracle_simulate(dict(cfg, ants={k: np.array([p[0], p[1], 0.0]) for k, p in cfg["ants"].items()}))
        assert rel_l2(flat, exact) > 100 * eps


def test_height_terms_at_decimetres_and_their_limit(gpu, monkeypatch):
    """Heights of a decimetre need more terms (the count follows (zh |s_z|)^K / K! <= eps / 10) and the 2-D plans
    run at eps / e^a; metres of scatter exceed 16 terms and take the 3-D transform."""
    from fftvis_amd.gpu import gpu_simulate

    c = synth.make_config("C2", nsrc=500, nfreq=2, ntimes=1)
    monkeypatch.setenv("FFTVIS_HIP_HANDLE_CACHE_BYTES", str(2**40))
    terms = []
    for sigma_m in (0.01, 0.1, 3.0):
        cfg = dict(c, ants=synth.with_z_scatter(c["ants"], sigma_m, seed=8))
        gpu_simulate.release_handles()
        got = fftvis_amd.simulate_vis(**cfg)
        terms.append(int(_last_handle_stats()["height_terms"]))
        assert rel_l2(got, oracle_simulate(cfg)) < TOL, sigma_m
    gpu_simulate.release_handles()
    assert 2 <= terms[0] < terms[1] <= 16 and terms[2] == 0, terms


def test_blocks_are_delivered_straight_into_a_slice_of_the_result(gpu, monkeypatch, tmp_path):
    """VERDICT r3 next #7 / DESIGN 9.4: ``simulate(out=vis[fsl, tsl])`` -- the reference's ``vis[tc][..., fc] = future``
    (cpu_simulate.py:846-847) without the copy.  A 78-MB block inside a 312-MB result (runs of 19.5 MB, one per
    channel, 58 MB apart): pinned run by run and filled from the copy stream (fv_sim_run_into), (a) in ordinary
    memory, (b) in a shared mapping another process could be writing to (the pinning helper only READS when it touches:
    a guard pattern around every run survives), (c) through the engine's own time blocks, (d) small blocks (2-D copy
    instead of pinning).  All equal the plain run of the same block bit for bit."""
    import mmap

    from fftvis_amd.gpu import gpu_simulate

    cfg = synth.make_config("C3", nsrc=2000, nfreq=8, ntimes=10)
    eng = fftvis_amd.create_simulation_engine("gpu")
    kw = dict(cfg, beam_list=[cfg["beam"]])
    kw.pop("beam")
    tsl, fsl = slice(2, 7), slice(2, 6)
    ref = eng.simulate(time_idx=tsl, freq_idx=fsl, **kw)
    assert ref.shape == (4, 5, 2, 2, 61075) and ref.nbytes > (64 << 20)
    shape = (8, 10, 2, 2, 61075)
    guard = 7.0 - 3.0j

    def check(full):
        got = full[fsl, tsl]
        assert np.array_equal(got, ref)
        mask = np.ones(shape[:2], bool)
        mask[fsl, tsl] = False
        assert np.all(full[mask] == guard)  # nothing outside the block was touched

    full = np.full(shape, guard, dtype=np.complex128)
    out = eng.simulate(time_idx=tsl, freq_idx=fsl, out=full[fsl, tsl], **kw)
    assert np.shares_memory(out, full)
    check(full)
    # (b) a shared mapping, as the ranks of a sharded run use it
    path = "/dev/shm/fftvis_amd_test_%d" % os.getpid()
    fd = os.open(path, os.O_CREAT | os.O_RDWR, 0o600)
    try:
        os.ftruncate(fd, int(np.prod(shape)) * 16)
        mm = mmap.mmap(fd, int(np.prod(shape)) * 16)
    finally:
        os.close(fd)
        os.unlink(path)
    shared = np.frombuffer(mm, dtype=np.complex128).reshape(shape)
    shared[...] = guard
    eng.simulate(time_idx=tsl, freq_idx=fsl, out=shared[fsl, tsl], out_shared=True, **kw)
    check(shared)
    # (c) the engine's own time blocks land in their slices of the caller's array
    full[...] = guard
    monkeypatch.setattr(gpu_simulate, "_time_block", lambda *a, **k: 2)
    eng.simulate(time_idx=tsl, freq_idx=fsl, out=full[fsl, tsl], **kw)
    monkeypatch.undo()
    check(full)
    # (d) a block too small to pin
    small = synth.make_config("C2", nsrc=300, nfreq=6, ntimes=5)
    ks = dict(small, beam_list=[small["beam"]])
    ks.pop("beam")
    rs = eng.simulate(time_idx=slice(1, 4), freq_idx=slice(2, 5), **ks)
    fs = np.full((6, 5) + rs.shape[2:], guard, dtype=np.complex128)
    eng.simulate(time_idx=slice(1, 4), freq_idx=slice(2, 5), out=fs[2:5, 1:4], **ks)
    assert np.array_equal(fs[2:5, 1:4], rs) and np.all(fs[:2] == guard) and np.all(fs[:, 4:] == guard)
    with pytest.raises(ValueError, match="out must be"):
        eng.simulate(time_idx=slice(1, 4), freq_idx=slice(2, 5), out=fs[2:4, 1:4], **ks)
    del shared
    mm.close()


def test_direct_third_dimension_equals_the_three_pass_transform(gpu, monkeypatch):
    """The 3-D transform (arrays whose heights are too large for the height-term expansion) runs x- and y-passes per
    (transform, z) plane and every target sums the planes with their exact phases; FFTVIS_HIP_NO_ZDIRECT=1 keeps the
    z-pass + 3-D gather.  Both against the oracle's exact sums and each other: HERA-7 with 1.5 m of scatter (small grids),
    HERA-350 with 1 m (8192^2-class planes: blocked B, column plan, source disc on every plane), packed and unpacked."""
    from fftvis_amd.gpu import gpu_simulate

    c1 = synth.make_config("C1")
    hrng = np.random.default_rng(5)
    rough = {k: np.array([v[0], v[1], 1.5 * hrng.normal()]) for k, v in c1["ants"].items()}
    tab = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(c1["freqs"]), c1["freqs"])
    big = _top_of_band(synth.make_config("C3", nsrc=20_000, ntimes=1, z_scatter=1.0), 2)
    cases = {
        "hera7 unpolarized": (dict(c1, ants=rough), None),
        "hera7 polarized table": (dict(c1, ants=rough, polarized=True, beam=tab), None),
        "hera350 1 m": (big, sorted(np.random.default_rng(2).choice(61075, 24, replace=False))),
    }
    monkeypatch.setenv("FFTVIS_HIP_HANDLE_CACHE_BYTES", str(2**40))
    for name, (cfg, sub) in cases.items():
        got = {}
        for mode in ("direct", "three-pass"):
            gpu_simulate.release_handles()
            if mode == "three-pass":
                monkeypatch.setenv("FFTVIS_HIP_NO_ZDIRECT", "1")
            got[mode] = fftvis_amd.simulate_vis(**cfg)
            st = _last_handle_stats()
            assert st["height_terms"] == 0 and st["n2_3"] > 1, (name, mode, st)  # a 3-D run, not the expansion
            monkeypatch.delenv("FFTVIS_HIP_NO_ZDIRECT", raising=False)
        gpu_simulate.release_handles()
        sel = cfg if sub is None else dict(cfg, baselines=[cfg["baselines"][i] for i in sub])
        exact = oracle_simulate(sel)
        for mode, v in got.items():
            vs = v if sub is None else v[..., sub]
            assert rel_l2(vs, exact) < TOL, (name, mode)
        assert rel_l2(got["direct"], got["three-pass"]) < 2 * TOL, name


def test_fuzz_heights_from_millimetres_to_metres(gpu, monkeypatch):
    """Seeded fuzz of the whole engine against the oracle with the antennas' height scatter drawn log-uniformly from
    1 mm to 3 m: height terms at every count K = 2 ... 16 and the direct 3-D transform beyond, random arrays, bands,
    catalogs, polarized skies, 1-3 beams with random assignment or eigenbeam coefficients, flipped pairs and autos,
    source chunks, both upsampling factors, eps in [1e-11, 1e-4].  (4 900 configurations of this generator, a
    quarter of them fp32, ran clean while writing it -- scratch/fuzz_r4.py; the two fp32 outliers were the float32
    rounding of 600-m baselines at 250 MHz, identical in all three paths.)"""
    from fftvis_amd.gpu import gpu_simulate

    monkeypatch.setenv("FFTVIS_HIP_HANDLE_CACHE_BYTES", str(2**40))
    # the light terms' second plan (looser tolerance, upsampling 1.25) is for large grids: here on these small ones too
    monkeypatch.setenv("FFTVIS_HIP_WTERM_LIGHT_CELLS", "0")
    rng = np.random.default_rng(404)
    kinds = set()
    light = 0
    for it in range(64):
        nant = int(rng.integers(4, 12))
        ext = float(np.exp(rng.uniform(np.log(10), np.log(400))))
        zs = float(np.exp(rng.uniform(np.log(1e-3), np.log(3.0))))
        ants = {i: np.array([rng.uniform(-ext, ext), rng.uniform(-ext, ext), rng.normal() * zs]) for i in range(nant)}
        nsrc, nfreq, ntimes = int(rng.integers(30, 300)), int(rng.integers(1, 5)), int(rng.integers(1, 4))
        freqs = np.sort(rng.uniform(50e6, 200e6) * (1 + rng.uniform(0, 0.4, nfreq)))
        times = np.linspace(2459845.0, 2459845.0 + rng.uniform(0.001, 0.3), ntimes)
        pol = bool(rng.uniform() < 0.6)
        ra, dec, flux = synth.catalog(nsrc, freqs, int(rng.integers(1e6)), polarized_sky=pol and rng.uniform() < 0.4)
        nbeam = int(rng.integers(1, 4))
        beams = [fftvis_amd.AiryBeam(float(rng.uniform(6, 16))) if rng.uniform() < 0.5 else
                 fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, float(rng.uniform(8, 15)), nza=46, naz=90), freqs)
                 for _ in range(nbeam)]
        allb = [(i, j) for i in range(nant) for j in range(nant)]
        sel = rng.choice(len(allb), size=min(len(allb), int(rng.integers(1, 40))), replace=False)
        eps = float(10 ** rng.uniform(-11, -4))
        cfg = dict(ants=ants, fluxes=flux, ra=ra, dec=dec, freqs=freqs, times=times, beam=beams if nbeam > 1 else beams[0],
                   beam_idx=rng.integers(0, nbeam, nant) if nbeam > 1 else None,
                   telescope_loc=(synth.HERA_LAT, synth.HERA_LON), baselines=[allb[k] for k in sel], polarized=pol,
                   precision=2, eps=eps, force_use_type3=True, coord_method="SiderealRotation")
        if pol and nbeam > 1 and rng.uniform() < 0.3:
            cfg["beam_coefs"] = rng.normal(size=(nant, nbeam, nfreq)) + 1j * rng.normal(size=(nant, nbeam, nfreq))
            cfg["beam_idx"] = None
        if rng.uniform() < 0.3:
            cfg["min_chunks"] = int(rng.integers(2, 4))
        if rng.uniform() < 0.25 and eps >= 1e-8:
            cfg["upsample_factor"] = 1.25
        gpu_simulate.release_handles()
        got = fftvis_amd.simulate_vis(**cfg)
        st = _last_handle_stats()
        kinds.add("K" if st["height_terms"] else "3-D" if st["n2_3"] > 1 else "2-D")
        light += st["height_terms_light_from"] > 0
        err = rel_l2(got, oracle_simulate(cfg))
        assert err < 10 * eps + 1e-12, (it, err, eps, zs, st["height_terms"], st["height_terms_light_from"], pol, nbeam)
    gpu_simulate.release_handles()
    assert kinds == {"K", "3-D", "2-D"} or kinds == {"K", "3-D"}, kinds
    assert light >= 10, light  # (18 000 configurations of scratch/fuzz_r4.py with this switch: half took the light plan, 2 beyond 10 eps -- a plain 2-D run and an fp32 one)


def test_light_height_terms_on_a_large_grid(gpu, monkeypatch):
    """The higher height terms enter with weights 2 |J_k(a)| -- 0.03 for k = 2, 8e-5 for k = 4 at 3 cm of scatter -- and
    run on a second plan at a looser tolerance and upsampling factor 1.25 (`Sim::wt_k0`): HERA-350 with 3 cm of height
    scatter takes it by itself (the grid is large), agrees with the run that keeps every term on the full plan far
    inside the tolerance, and both agree with the oracle's exact 3-D sums on a subset of the baselines."""
    from fftvis_amd.gpu import gpu_simulate

    cfg = synth.make_config("C3", nsrc=3000, nfreq=2, ntimes=1, z_scatter=0.03)
    sub = sorted(np.random.default_rng(1).choice(61075, 64, replace=False))
    gpu_simulate.release_handles()
    got = fftvis_amd.simulate_vis(**cfg)
    st = _last_handle_stats()
    assert st["height_terms"] >= 5 and 1 <= st["height_terms_light_from"] <= st["height_terms"] - 2, st
    assert st["height_terms_light_from"] < st["height_terms_lighter_from"] <= st["height_terms"] - 2, st  # two light classes here
    exp = oracle_simulate(dict(cfg, baselines=[cfg["baselines"][i] for i in sub]))
    assert rel_l2(got[..., sub], exp) < TOL
    monkeypatch.setenv("FFTVIS_HIP_NO_WTERM_LIGHT", "1")
    gpu_simulate.release_handles()
    full = fftvis_amd.simulate_vis(**cfg)
    assert _last_handle_stats()["height_terms_light_from"] == 0
    gpu_simulate.release_handles()
    assert rel_l2(got, full) < 0.3 * cfg["eps"]
    monkeypatch.delenv("FFTVIS_HIP_NO_WTERM_LIGHT")
    monkeypatch.setenv("FFTVIS_HIP_WTERM_ONE_LIGHT_CLASS", "1")
    one = fftvis_amd.simulate_vis(**cfg)
    st1 = _last_handle_stats()
    assert st1["height_terms_light_from"] > 0 and st1["height_terms_lighter_from"] == 0
    gpu_simulate.release_handles()
    assert rel_l2(one, full) < 0.3 * cfg["eps"]


@pytest.mark.parametrize("order", [0, 2, 4, 5])
def test_beam_spline_orders_other_than_1_and_3(gpu, order):
    """beam_spline_opts {"order": n} for the other orders scipy.ndimage.map_coordinates takes (the reference hands the
    option to pyuvdata's az_za_map_coordinates, cpu/beams.py:69-74): the device's general path -- prefilter with
    that order's poles, n + 1 nodes per axis weighted by the cardinal B-spline -- against the oracle (scipy's own
    spline_filter1d + the closed piecewise polynomials), stand-alone and through a polarized simulation on both
    engine paths."""
    freqs = np.linspace(100e6, 120e6, 4)
    tab = fftvis_amd.TabulatedBeam(synth.synthetic_efield_table(freqs, nza=46, naz=90), freqs)
    rng = np.random.default_rng(order)
    az, za = rng.uniform(0, 2 * np.pi, 3000), rng.uniform(0, np.pi / 2, 3000)
    za[:3] = [0.0, np.pi / 2, 1e-12]
    az[:3] = [0.0, 2 * np.pi - 1e-12, np.pi]
    from fftvis_amd.gpu import GPUBeamEvaluator

    ev = GPUBeamEvaluator()
    opts = {"order": order}
    for pol in (False, True):
        for fi in (0, 3):
            got = ev.evaluate_beam(tab, az, za, pol, freqs[fi], freq_index=fi, spline_opts=opts)
            exp = orc.evaluate_beam(oracle_beam(tab, pol, freqs, order), az, za, pol, freqs[fi])
            if order == 0:  # nearest node: a point within rounding of a cell edge may take either neighbour
                frac = np.minimum(np.abs(np.mod(az / (2 * np.pi / 90), 1) - 0.5), np.abs(np.mod(za / (np.pi / 45), 1) - 0.5))
                keep = frac > 1e-9
                got, exp = got[..., keep], exp[..., keep]
            np.testing.assert_allclose(got, exp, rtol=1e-11, atol=1e-13)
    cfg = synth.make_config("C1", nsrc=300, nfreq=4, ntimes=2)
    cfg = dict(cfg, freqs=freqs, fluxes=cfg["fluxes"][:, :4], beam=tab, polarized=True, eps=1e-10, beam_spline_opts=opts)
    for path in (True, False):
        c = dict(cfg, force_use_type3=path)
        got = fftvis_amd.simulate_vis(**c)
        exp = oracle_simulate(c)
        assert rel_l2(got, exp) < 2e-9, (order, path, rel_l2(got, exp))
    if order in (2, 5):  # and it IS a different interpolant from its neighbours
        other = fftvis_amd.simulate_vis(**dict(cfg, beam_spline_opts={"order": 3}))
        assert rel_l2(other, exp) > 1e-7


def test_fp32_low_upsampling_does_not_degrade_below_its_floor(gpu):
    """fp32 at upsample_factor 1.25: asking for more than the combination can deliver must not make the result WORSE.
    Beyond ~10 cells (8 in 3-D) a wider kernel only amplifies fp32 rounding at band-edge targets (Nufft3's cap):
    before it, the fp32 default tolerance 6e-8 gave 4.6e-5 (2-D) / 5.7e-4 (3-D) where eps = 1e-4 gives 3.5e-6."""
    import warnings

    for cfg, floor in ((synth.make_config("C1", nsrc=300, nfreq=3, ntimes=2), 1e-5),
                       (synth.make_config("C1", nsrc=300, nfreq=3, ntimes=2, z_scatter=2.0), 1e-4)):
        exp = oracle_simulate(dict(cfg, precision=2))
        errs = {}
        for eps in (1e-4, 6e-8, 1e-9):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)  # (the engine says so itself: test_upsample_1p25_...)
                got = fftvis_amd.simulate_vis(**dict(cfg, precision=1, upsample_factor=1.25, eps=eps))
            errs[eps] = rel_l2(got, exp)
        assert errs[6e-8] < floor and errs[1e-9] < floor, errs
        assert errs[1e-9] < 3 * errs[1e-4] + 1e-6, errs


def test_fuzz_with_every_spline_order(gpu):
    """Seeded fuzz of the whole engine against the oracle as in test_sim_fuzz_random_configurations, with the table beams'
    interpolation order drawn from 0 .. 5, every seventh configuration at upsample_factor 1.25 and every eleventh in
    fp32 (at the documented floors of that combination).  (1 800 configurations of the same generator, seeds 7000 /
    8000 / 9000, ran clean while writing it -- scratch/fuzz_final.py: worst error 7.3 eps; the one outlier they had
    found is the fp32 cap above.)"""
    import warnings

    from tests.test_gpu_parity import _random_sim_config

    rng = np.random.default_rng(7000)
    for it in range(72):
        cfg = _random_sim_config(rng, lattice=it % 5 == 4)
        beams = cfg["beam"] if isinstance(cfg["beam"], list) else [cfg["beam"]]
        if any(isinstance(b, fftvis_amd.TabulatedBeam) for b in beams):
            cfg["beam_spline_opts"] = {"order": int(rng.integers(0, 6))}
        if it % 7 == 3:
            cfg.update(upsample_factor=1.25, eps=max(cfg["eps"], 1e-8))
        if it % 11 == 5:
            cfg.update(precision=1, eps=max(cfg["eps"], 2e-5))
            if cfg.get("upsample_factor") == 1.25:
                flat = np.ptp([p[2] for p in cfg["ants"].values()]) < 1e-3
                cfg["eps"] = max(cfg["eps"], 1e-4 if flat else 1e-3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            err = rel_l2(fftvis_amd.simulate_vis(**cfg), oracle_simulate(cfg))
        lim = 10 * cfg["eps"] + (1e-12 if cfg.get("precision", 2) == 2 else 2e-6)
        assert err < lim, (it, err, cfg["eps"], cfg.get("beam_spline_opts"), cfg.get("precision", 2), cfg.get("upsample_factor"))


def test_c_abi_catalog_exchange_over_rccl_one_rank(gpu):
    """The optional RCCL entry points of the C ABI (SURVEY 8 b: fv_comm_init / fv_bcast_catalog; 8 e: only the
    frequency columns a rank needs), as far as one GPU can exercise them: a communicator of ONE rank, the broadcast
    in place, and the packed column transfer (root = the only rank: its send meets its own receive inside one group)
    for real and for 2 x 2 complex flux entries -- the columns must arrive bit for bit.  More ranks need more GPUs:
    the multi-process tests cover the torch.distributed twin (parallel.broadcast_catalog_device)."""
    import ctypes

    import torch

    from fftvis_amd import _lib

    L = _lib.lib()
    ident = (ctypes.c_ubyte * 128)()
    _lib.check(L.fv_comm_unique_id(ident))
    assert any(ident)
    comm = ctypes.c_void_p()
    _lib.check(L.fv_comm_init(ctypes.byref(comm), 0, 0, 1, ident))
    try:
        g = torch.Generator(device="cpu").manual_seed(5)
        nsrc, nfreq = 1000, 12
        eq = torch.randn(3, nsrc, dtype=torch.float64, generator=g).cuda()
        for shape, dtype in (((nsrc, nfreq), torch.float64), ((nsrc, nfreq), torch.float32), ((nsrc, nfreq, 2, 2), torch.complex128)):
            flux = torch.randn(*shape, dtype=torch.float64, generator=g).to(dtype).cuda()
            if dtype == torch.complex128:
                flux = flux + 1j * torch.randn(*shape, dtype=torch.float64, generator=g).cuda()
            keep_eq, keep_flux = eq.clone(), flux.clone()
            _lib.check(L.fv_bcast_catalog(comm, 0, eq.data_ptr(), eq.numel() * eq.element_size(), flux.data_ptr(),
                                          flux.numel() * flux.element_size()))
            assert torch.equal(eq, keep_eq) and torch.equal(flux, keep_flux)
            f0, f1 = 3, 9
            out = torch.zeros((nsrc, f1 - f0) + tuple(shape[2:]), dtype=dtype, device="cuda")
            ranges = (ctypes.c_int * 2)(f0, f1)
            elem = flux.element_size() * (4 if len(shape) == 4 else 1)
            _lib.check(L.fv_scatter_flux_columns(comm, 0, nsrc, nfreq, elem, flux.data_ptr(), ranges, out.data_ptr()))
            assert torch.equal(out, flux[:, f0:f1])
        # arguments are checked before anything is sent
        bad = (ctypes.c_int * 2)(5, 20)
        assert L.fv_scatter_flux_columns(comm, 0, nsrc, nfreq, 8, eq.data_ptr(), bad, eq.data_ptr()) == 1  # FV_ERR_ARG
        assert L.fv_bcast_catalog(comm, 3, eq.data_ptr(), 8, None, 0) == 1
    finally:
        _lib.check(L.fv_comm_destroy(comm))


def test_either_import_order_leaves_one_hip_runtime(gpu):
    """The library first, PyTorch afterwards (a host that only later reaches for torch.distributed): PyTorch-ROCm wheels
    bring their own HIP runtime, and two of them in one process leave the second without a GPU.  ``_lib.lib()`` loads
    the wheel's copy before the library, so that both bind to one runtime whichever comes first."""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from fftvis_amd import _lib\n"
        "assert 'torch' not in sys.modules\n"
        "assert _lib.device_count() >= 1\n"
        "from fftvis_amd.gpu import gpu_nufft2d\n"
        "x = np.linspace(-1, 1, 50); c = np.ones(50, complex)\n"
        "v = gpu_nufft2d(x, x, c, np.array([0.0, 1.0]), np.array([0.0, 2.0]), 1e-9)\n"
        "assert abs(v.ravel()[0] - 50) < 1e-6\n"
        "import torch\n"
        "assert torch.cuda.is_available(), 'a second HIP runtime came in with torch'\n"
        "assert float(torch.ones(4, device='cuda').sum()) == 4.0\n"
        "maps = open('/proc/self/maps').read()\n"
        "libs = {l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}\n"
        "assert len(libs) == 1, libs\n"
        "print('ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_integration_stub_call_sequence_with_raw_ctypes(gpu):
    """INTEGRATION.md's stub, run as written -- `ctypes.CDLL` on the library, no `fftvis_amd._lib` / `SimHandle` in
    between: create, sources, per-time rotations, frequencies, array, one Airy beam, one pair list, chunking, run, destroy
    -- against the oracle (unpolarized C1)."""
    import ctypes

    from fftvis_amd import _lib
    from fftvis_amd.core import coords, utils
    from fftvis_amd.gpu.gpu_simulate import prepare_array

    cfg = synth.make_config("C1", nsrc=200, nfreq=4, ntimes=2)
    _lib.lib()  # (the library is built, and the wheel's HIP runtime is in: what an embedding host does once)
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.fv_last_error.restype = ctypes.c_char_p

    def ck(st):
        if st:
            raise RuntimeError(L.fv_last_error().decode())

    def vp(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    eps, precision, polarized = 1e-9, 2, False
    freqs64 = np.ascontiguousarray(cfg["freqs"], dtype=np.float64)
    nsrc, nfreqs, ntimes = len(cfg["ra"]), len(freqs64), len(cfg["times"])
    eq_xyz = np.ascontiguousarray(coords.eq_unit_vectors(cfg["ra"], cfg["dec"]))
    coherency = np.ascontiguousarray(utils.prepare_source_catalog(cfg["fluxes"], polarized)[0], dtype=np.float64)
    rot = np.ascontiguousarray(coords.SiderealRotation(cfg["times"], cfg["telescope_loc"]).matrices())
    R, bls, is_coplanar = prepare_array(cfg["ants"], cfg["baselines"], 1e-6, np.float64)
    nbls = bls.shape[1]
    rotation_matrix64, bls_seconds64 = np.ascontiguousarray(R, dtype=np.float64), np.ascontiguousarray(bls, dtype=np.float64)
    bi, bj = np.zeros(1, np.int32), np.zeros(1, np.int32)
    offsets = np.array([0, nbls], dtype=np.int64)
    bl_idx, flipped = np.arange(nbls, dtype=np.int32), np.zeros(nbls, np.int8)

    h = ctypes.c_void_p()
    ck(L.fv_sim_create(ctypes.byref(h), 0, precision, ctypes.c_double(eps), ctypes.c_double(2.0), int(polarized)))
    try:
        ck(L.fv_sim_set_sources(h, ctypes.c_int64(nsrc), nfreqs, vp(eq_xyz), vp(coherency), 0, 0))
        ck(L.fv_sim_set_times(h, ntimes, vp(rot)))
        ck(L.fv_sim_set_freqs(h, nfreqs, vp(freqs64)))
        ck(L.fv_sim_set_array(h, vp(rotation_matrix64), ctypes.c_int64(nbls), vp(bls_seconds64), int(is_coplanar)))
        ck(L.fv_sim_set_nbeams(h, 1))
        ck(L.fv_sim_set_beam_airy(h, 0, ctypes.c_double(cfg["beam"].diameter)))
        ck(L.fv_sim_set_beam_pairs(h, 1, vp(bi), vp(bj), vp(offsets), vp(bl_idx), vp(flipped)))
        ck(L.fv_sim_set_chunking(h, 1, ctypes.c_double(1.0)))
        vis = np.empty((nfreqs, ntimes, nbls), np.complex128)
        ck(L.fv_sim_run(h, 0, ntimes, 0, nfreqs, vp(vis), 0))
    finally:
        L.fv_sim_destroy(h)
    assert rel_l2(vis, oracle_simulate(dict(cfg, eps=eps))) < 5 * eps
