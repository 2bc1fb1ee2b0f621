#!/usr/bin/env python3
"""bench.py -- simulated visibilities/s of the fftvis hot path on MI355X.

A "step" is one pass of the hot path (rotate -> beam -> coherency -> type-3 NUFFT for every
(time, frequency) slice) over one BASELINE.json configuration with seeded synthetic inputs that
are already resident in HBM when the timed region starts.  Default workload: configs[2]
("C3": HERA-350, 1e5 sources, 128 freqs, 20 times, polarized table beam, fp64, eps = 6e-8,
upsample_factor 2) -- the largest configuration BASELINE.json lists for one GPU.  `--workload C2`
keeps the HERA-37 case of round 1, C4 / C5 run the 8-GPU configurations' shapes.

Multi-GPU (--gpus N; one rank per GPU, either launched by torch.distributed.run or, when bench.py is
started as a single process, by bench.py itself as a child torchrun -- spawn_ranks): ONE observation is
sharded over the ranks by independent (time, frequency) blocks (parallel.shard_blocks_weighted:
time-major, frequency cuts balanced by the nu^2 grid cost) with no data-path collective; the only
communication is the one-off RCCL broadcast of the source catalog from rank 0 into every rank's
HBM before the timed region, plus the barriers / MAX-reduce of the timing contract.  Total work
is fixed as N grows: "scaling": "strong".

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def pmc_file():
    """The newest committed PMC summary (profiles/rNN_hbm_traffic_pmc.json, tools/refresh_profiles.py)."""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic_pmc.json")))
    return os.path.relpath(found[-1], ROOT) if found else None


def git_blob_id(path):
    """`git hash-object` of a file, computed here (the GPU box has no .git): names the exact committed content."""
    import hashlib

    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None, help="timed steps (default: 3; 50 for C2)")
    p.add_argument("--warmup", type=int, default=None,
                   help="untimed steps first (default: 1; 20 for C2 -- a 1.5 ms step needs ~30 ms of work "
                        "before clocks and caches settle)")
    p.add_argument("--workload", default=os.environ.get("FFTVIS_BENCH_WORKLOAD", "C3"),
                   choices=["C1", "C2", "C3", "C3z", "C4", "C5"],
                   help="C3z = C3's array with a seeded 3 cm height scatter (non-coplanar: the 3-D transform the "
                        "reference takes whenever any |b_z| > 1e-6 m, cpu_simulate.py:655), 8 channels x 2 times")
    p.add_argument("--array", default=None, choices=["scattered"],
                   help="scattered = 350 antennas uniformly inside HERA-350's footprint: no lattice, no repeated "
                        "baseline vectors (the generic type-3 workload; same catalog / beam / band)")
    p.add_argument("--z-scatter", type=float, default=None, help="height scatter in metres (C3z: 0.03)")
    p.add_argument("--no-extras", action="store_true",
                   help="skip the short side runs of the default line (generic_array, default_path_gpu, c2_ms_per_step)")
    p.add_argument("--nsrc", type=int, default=None)
    p.add_argument("--nfreq", type=int, default=None)
    p.add_argument("--ntimes", type=int, default=None)
    p.add_argument("--eps", type=float, default=None, help="default: the workload's (6e-8; C5: 1e-4)")
    p.add_argument("--upsample", type=float, default=2.0,
                   help="2 (the reference's default upsample_factor), 1.25, or 0 = the engine picks per run")
    p.add_argument("--path", default="type3", choices=["type3", "type1"],
                   help="type3 = the benchmarked NUFFT path (BASELINE.json); type1 = the lattice path "
                        "the reference takes by default on these arrays (reported for comparison)")
    p.add_argument("--lanes", type=int, default=None, choices=[1, 2], help="FFTVIS_HIP_LANES (small grids)")
    p.add_argument("--pipe", type=int, default=None, choices=[0, 1], help="FFTVIS_HIP_PIPE (small grids)")
    p.add_argument("--as-rank", type=int, default=None,
                   help="with --of-ranks N: one process runs ONLY the block rank R of an N-rank job would own "
                        "(per-rank cost of the sharding without N GPUs; value = that block's visibilities/s)")
    p.add_argument("--of-ranks", type=int, default=None)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-e2e", action="store_true",
                   help="skip the host-in -> host-out simulate_vis() calls of the `e2e` entry")
    p.add_argument("--no-breakdown", action="store_true",
                   help="skip the extra single-stream step that times every kernel family (profiling runs)")
    p.add_argument("--cpu-seconds", type=float, default=20.0)
    a = p.parse_args()
    a.base = "C3" if a.workload == "C3z" else a.workload
    if a.workload == "C3z":
        a.nfreq = a.nfreq or 8
        a.ntimes = a.ntimes or 2
        a.z_scatter = 0.03 if a.z_scatter is None else a.z_scatter
    a.z_scatter = a.z_scatter or 0.0
    a.light = a.workload in ("C1", "C2") and not (a.nsrc and a.nsrc > 200000)
    if a.steps is None:
        a.steps = 50 if a.light else 3
    if a.warmup is None:
        a.warmup = 20 if a.light else 1
    return a


def cpu_baseline(cfg, seconds: float):
    """The CPU path timed beside the GPU on a bounded sample of (time, frequency) slices, one NUFFT call per
    slice and beam pair as the reference does (cpu_simulate.py:969-1069).  Slices are taken from the top of the
    band downwards and the bottom upwards alternately so that the sample's mean cost is the band's.

    If ``finufft`` is importable on this box the NUFFT is the reference's own call (cpu/nufft.py:48-59:
    ``finufft.nufft2d3(x, y, c, u, v, modeord=0, eps=eps, nthreads=all cores, showwarn=0, upsampfac=2)``) --
    ``kind: "finufft"`` -- and a second, shorter sample times the type-1 call the reference takes by default on
    these lattice arrays (cpu/nufft.py:162-175) for the ``default_path`` entry.  Otherwise the NUFFT is the
    build's own CPU port (oracle/cpu_nufft: C/OpenMP spread + gather around scipy.fft, all cores) --
    ``kind: "port"``, which is NOT finufft.  Beam and coherency are the oracle's numpy restatement either way."""
    from oracle import cpu_nufft
    from oracle import fftvis_oracle as orc
    from tests.helpers import oracle_beam

    try:
        ncores = len(os.sched_getaffinity(0))  # the cores this process may run on
    except AttributeError:
        ncores = os.cpu_count()
    try:
        import finufft
    except Exception:
        finufft = None
    eps = cfg["eps"]
    freqs, times = cfg["freqs"], cfg["times"]
    pol = cfg["polarized"]
    nfeeds = 2 if pol else 1
    ants = cfg["ants"]
    key2idx = {a: i for i, a in enumerate(ants)}
    antvecs = np.array([ants[a] for a in ants], dtype=float)
    bls = np.array([antvecs[key2idx[b[1]]] - antvecs[key2idx[b[0]]] for b in cfg["baselines"]]).T
    bls = bls / orc.speed_of_light
    coh, pol_sky = orc.prepare_source_catalog(cfg["fluxes"], pol)
    mgr = orc.SimpleCoordinateRotation(coh, times, cfg["telescope_loc"], cfg["ra"], cfg["dec"])
    blist = cfg["beam"] if isinstance(cfg["beam"], list) else [cfg["beam"]]
    beams = [oracle_beam(b, pol, freqs) for b in blist]
    # eigenbeam workloads: one NUFFT per basis pair k <= l (reference cpu_simulate.py:416-417)
    bpairs = [(k, l) for k in range(len(beams)) for l in range(k, len(beams))] if "beam_coefs" in cfg else [(0, 0)]
    nf = len(freqs)
    order = [i // 2 if i % 2 == 0 else nf - 1 - i // 2 for i in range(nf)]  # bottom, top, bottom + 1, ...

    def type3(topo, c, uvw, fi):
        if finufft is not None:
            return finufft.nufft2d3(topo[0], topo[1], np.ascontiguousarray(c), np.ascontiguousarray(uvw[0]),
                                    np.ascontiguousarray(uvw[1]), modeord=0, eps=eps, nthreads=ncores, showwarn=0,
                                    upsampfac=2)
        return cpu_nufft.nufft_type3([topo[0], topo[1]], c, [uvw[0], uvw[1]], eps=eps)

    def sample(nufft, budget, prepare=None):
        nslices, t_used, n = 0, 0.0, 0
        t_start = time.perf_counter()
        for ti in range(len(times)):
            mgr.rotate(ti)
            topo, flux, n = mgr.select_chunk(0, ti)
            az, za = orc.enu_to_az_za(topo[0], topo[1])
            topo = 2 * np.pi * (prepare(topo) if prepare else topo)
            for fi in order:
                bev = [orc.evaluate_beam(b, az, za, pol, freqs[fi]).astype(complex) for b in beams]
                uvw = bls * freqs[fi]
                for (k, l) in bpairs:
                    c = orc.compute_apparent_coherency(bev, k, l, flux, fi, pol, pol_sky, nfeeds)
                    nufft(topo, c.reshape(-1, c.shape[-1]) if c.ndim > 1 else c, uvw, fi)
                nslices += 1
                t_used = time.perf_counter() - t_start
                if t_used > budget and nslices % 2 == 0:
                    break
            if t_used > budget:
                break
        return nslices, t_used, n

    nslices, t_used, n = sample(type3, seconds)
    nbls = len(cfg["baselines"])
    w = cpu_nufft.es_params(eps, 2.0)[0]
    nfe = (nfeeds * nfeeds)
    res = {
        "value": nbls * nslices / t_used,
        "unit": "visibilities/s",
        "cores": ncores,
        "kind": "finufft" if finufft is not None else "port",
    }
    if finufft is not None:
        res["sample"] = (f"{nslices} (time,freq) slices of the workload (alternating from both ends of the band) in "
                         f"{t_used:.1f} s; finufft {getattr(finufft, '__version__', '?')} nufft2d3, eps={eps:g}, "
                         f"upsampfac=2, nthreads={ncores} -- the reference's own type-3 call (cpu/nufft.py:48-59); "
                         "numpy beam/coherency")
        # the path the reference takes BY DEFAULT on these lattice arrays (cpu_simulate.py:634-637, 661-681):
        # type 1 onto the lattice's mode grid, then the baselines' modes are picked (cpu/nufft.py:120-175)
        try:
            from fftvis_amd.core.antenna_gridding import check_antpos_griddability

            ok, grid, basis = check_antpos_griddability(ants)
            if ok and "beam_coefs" not in cfg:
                bint = np.round(np.array([grid[b[1]] - grid[b[0]] for b in cfg["baselines"]]).T).astype(int)
                n_modes = 2 * int(np.abs(bint).max()) + 1
                B = basis / orc.speed_of_light

                def type1(topo, c, uvw, fi):
                    model = finufft.nufft2d1(topo[0] * freqs[fi], topo[1] * freqs[fi], np.ascontiguousarray(c),
                                             n_modes, modeord=1, eps=eps, nthreads=ncores, showwarn=0, upsampfac=2)
                    return model[..., bint[0], bint[1]]

                ns1, t1, _ = sample(type1, max(4.0, seconds / 4), prepare=lambda topo: B.T @ topo)
                res["default_path"] = {
                    "value": nbls * ns1 / t1, "unit": "visibilities/s", "kind": "finufft type 1",
                    "sample": f"{ns1} slices in {t1:.1f} s; finufft.nufft2d1 on the {n_modes}^2 lattice mode grid + "
                              "mode pick (cpu/nufft.py:162-175): what simulate_vis runs on this array unless "
                              "force_use_type3",
                }
        except Exception as e:  # the like-for-like number stands on its own
            res["default_path"] = {"error": repr(e)}
    else:
        thr_spread = cpu_nufft._nthreads(n * nfe * w * w)
        thr_interp = cpu_nufft._nthreads(nbls * nfe * w * w)
        res["sample"] = (f"{nslices} (time,freq) slices of the workload (alternating from both ends of the band) in "
                         f"{t_used:.1f} s; scipy.fft with {ncores} workers, C/OpenMP spread / interp (oracle/cpu_nufft.c) "
                         f"with {thr_spread} / {thr_interp} threads, numpy beam/coherency -- CPU restatement of "
                         "the type-3 NUFFT path, not finufft (finufft is not importable on this box)")
    return res


def spawn_ranks(n: int) -> int:
    """`bench.py --gpus N` started as ONE process (no WORLD_SIZE in the environment): start the N ranks
    ourselves, the way the reference fans its own workers out of one call (cpu_simulate.py:714-835) -- a
    CHILD `python -m torch.distributed.run` (never an exec, and before this process touches the GPU), one
    rank per GPU, rendezvous on 127.0.0.1.  Rank 0's JSON line passes through on stdout; returns the
    child's exit code."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free port for the rendezvous
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only mode the host driver supports
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' stdout carries rank 0's JSON line -- and, under gloo, the transport's connection chatter: only the JSON
    # line goes to our stdout (the contract is ONE line), everything else to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def device_catalog(cfg, dev):
    """The prepared catalog (unit vectors, coherency per source and channel) as device tensors."""
    import torch

    from fftvis_amd import parallel
    from fftvis_amd.core import utils
    from fftvis_amd.core.coords import eq_unit_vectors

    precision = cfg.get("precision", 2)
    coh, pol_sky = utils.prepare_source_catalog(cfg["fluxes"], cfg["polarized"])
    rdt = torch.float32 if precision == 1 else torch.float64
    cdt = torch.complex64 if precision == 1 else torch.complex128
    return parallel.DeviceCatalog(torch.from_numpy(eq_unit_vectors(cfg["ra"], cfg["dec"])).to(dev, rdt),
                                  torch.from_numpy(np.ascontiguousarray(coh)).to(dev, cdt if pol_sky else rdt), pol_sky)


def make_handle(cfg, cat, device_index, eps, upsample, path):
    """An engine handle configured for one workload (catalog already in HBM); returns (handle, coplanar)."""
    from fftvis_amd.core import utils
    from fftvis_amd.core.coords import SiderealRotation
    from fftvis_amd.gpu.gpu_simulate import SimHandle, prepare_array

    freqs, baselines = cfg["freqs"], cfg["baselines"]
    R, bls, coplanar = prepare_array(cfg["ants"], baselines, 1e-6, np.float64)
    pairs, pidx, pflip = utils.prepare_beam_evaluation(list(cfg["ants"]), baselines, None)
    h = SimHandle(device_index, cfg.get("precision", 2), eps, upsample, cfg["polarized"])
    h.set_sources_device(cat.nsrc, cat.nfreq, cat.eq.data_ptr(), cat.flux.data_ptr(), cat.polarized_sky)
    h.set_times(SiderealRotation(cfg["times"], cfg["telescope_loc"]).matrices())
    h.set_freqs(freqs)
    if path == "type1":
        from fftvis_amd.core.antenna_gridding import check_antpos_griddability

        ok, grid, basis = check_antpos_griddability(cfg["ants"])
        assert ok, "workload array is not a lattice"
        bint = np.round(np.array([grid[b[1]] - grid[b[0]] for b in baselines]).T).astype(int)
        h.set_array_type1(basis / utils.speed_of_light, bint, 2 * int(np.abs(bint).max()) + 1)
    else:
        h.set_array(R, bls, coplanar)
    blist = cfg["beam"] if isinstance(cfg["beam"], list) else [cfg["beam"]]
    h.set_beams(blist, freqs)
    if "beam_coefs" in cfg:  # eigenbeam workload (C5): K (K + 1) / 2 NUFFTs per slice
        antnums = list(cfg["ants"])
        h.set_basis(cfg["beam_coefs"], [antnums.index(b[0]) for b in baselines],
                    [antnums.index(b[1]) for b in baselines])
    else:
        h.set_beam_pairs(pairs, pidx, pflip)
    return h, coplanar


def mini_run(cfg, device_index, path="type3", upsample=2.0, steps=2, warmup=1, settle_s=0.0, breakdown=False):
    """A short device-resident run of ANOTHER workload on this GPU with its own handle -- the side entries of the
    default line (generic_array, default_path_gpu, c2_ms_per_step): same step definition, same timing bracket
    (synchronise, K steps, synchronise), inputs resident in HBM, output left on the device."""
    import torch

    dev = torch.device("cuda", device_index)
    cat = device_catalog(cfg, dev)
    h, coplanar = make_handle(cfg, cat, device_index, cfg["eps"], upsample, path)
    nt, nf = len(cfg["times"]), len(cfg["freqs"])
    cdt = torch.complex64 if cfg.get("precision", 2) == 1 else torch.complex128
    out = torch.empty(h.out_shape(nt, nf), dtype=cdt, device=dev)

    def step():
        h.run_device(0, nt, 0, nf, out.data_ptr())

    t_s = time.perf_counter()
    for _ in range(warmup):
        step()
    h.sync()
    while time.perf_counter() - t_s < settle_s:
        step()
        h.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    h.sync()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    res = {"ms_per_step": ms, "value": len(cfg["baselines"]) * nf * nt / (ms * 1e-3), "unit": "visibilities/s",
           "steps": steps, "warmup": warmup,
           "finite_output": bool(torch.isfinite(torch.view_as_real(out)).all().item())}
    if breakdown:
        h.reset_stats()
        h.enable_timing(2)
        step()
        h.sync()
        tm, st = h.timing(), h.stats()
        h.enable_timing(0)
        RB = 4.0 if cfg.get("precision", 2) == 1 else 8.0
        fam = ("spread", "fft", "interp", "strengths", "prep")
        total = max(sum(tm[k] for k in fam), 1e-9)
        l2 = max(st["spread_launches"], 1.0)
        res["share_of_step"] = {k: tm[k] / total for k in fam}
        res["launches_per_step"] = l2
        if tm["fft"] > 0:
            gbps = st["fft_cells"] * 2 * RB / (tm["fft"] * 1e-3) / 1e9
            res["roofline_fft"] = {"achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
                                   "fft_ms_per_launch": tm["fft"] / l2}
            if st.get("fft_flops", 0) > 0:
                peak_tf = 78.6 if RB == 8 else 157.3
                res["roofline_fft"]["flops_frac"] = st["fft_flops"] / (tm["fft"] * 1e-3) / 1e12 / peak_tf
        if tm["spread"] > 0:
            res["spread_ms_per_launch"] = tm["spread"] / l2
        if tm["interp"] > 0:
            res["interp_ms_per_launch"] = tm["interp"] / l2
        res["grid_top_channel"] = {"n2": [int(st["n2x"]), int(st["n2y"])],
                                   "active": [int(st["n2z"]) // 65536, int(st["n2z"]) % 65536]}
    h.close()
    del out, cat
    torch.cuda.empty_cache()
    return res


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and a.as_rank is None:
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and a.as_rank is None:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but {world} rank(s) joined (WORLD_SIZE); launch with "
                         f"--nproc-per-node {a.gpus} or let bench.py start its own ranks")
    import torch

    from fftvis_amd import _lib, parallel, synth

    if a.lanes is not None:
        os.environ["FFTVIS_HIP_LANES"] = str(a.lanes)
    if a.pipe is not None:
        os.environ["FFTVIS_HIP_PIPE"] = str(a.pipe)
    _lib.require_gpu()
    # rehearsal switches (a one-GPU box cannot host two RCCL ranks): FFTVIS_BENCH_BACKEND=gloo
    # with FFTVIS_BENCH_SHARE_GPU=1 runs the N > 1 code path with every rank on device 0
    backend = os.environ.get("FFTVIS_BENCH_BACKEND", "nccl")
    if os.environ.get("FFTVIS_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # FFTVIS_BENCH_FORCE_DIST=1: take the N > 1 code path with ONE rank -- RCCL itself, its process group, barriers,
    # object collectives and the sharded host-to-host calls as an 8-GPU node will run them, as far as one GPU can
    force_dist = world == 1 and os.environ.get("FFTVIS_BENCH_FORCE_DIST") == "1" and a.as_rank is None
    if force_dist:
        import socket

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
    if world > 1 or force_dist:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL on ROCm
        else:
            dist.init_process_group(backend)

    cfg = synth.make_config(a.base, nsrc=a.nsrc, nfreq=a.nfreq, ntimes=a.ntimes,
                            array="scattered350" if a.array == "scattered" else None, z_scatter=a.z_scatter)
    if a.eps is None:
        a.eps = cfg["eps"]
    cfg["eps"] = a.eps
    precision = cfg.get("precision", 2)
    cdt = torch.complex64 if precision == 1 else torch.complex128
    RB = 4.0 if precision == 1 else 8.0  # bytes per real
    freqs, pol = cfg["freqs"], cfg["polarized"]
    ntimes, nfreq = len(cfg["times"]), len(freqs)
    baselines = cfg["baselines"]
    nbls = len(baselines)
    nsrc = len(cfg["ra"])

    # ---- catalog: prepared on rank 0, broadcast over RCCL/xGMI into every rank's HBM ------------
    if dist is not None:
        cat = parallel.broadcast_catalog_device(cfg["ra"] if rank == 0 else None, cfg["dec"] if rank == 0 else None,
                                                cfg["fluxes"] if rank == 0 else None, pol, precision, dev,
                                                via_host=backend != "nccl")
    else:
        cat = device_catalog(cfg, dev)
    torch.cuda.synchronize()

    # ---- this rank's block of the observation ----------------------------------------------------
    blocks = parallel.shard_blocks_weighted(world, freqs, ntimes, nsrc)
    mine = blocks[rank]
    if a.as_rank is not None:
        assert world == 1 and a.of_ranks and 0 <= a.as_rank < a.of_ranks, "--as-rank R --of-ranks N, single process"
        blocks = [parallel.shard_blocks_weighted(a.of_ranks, freqs, ntimes, nsrc)[a.as_rank]]
        mine = blocks[0]
    h, coplanar = make_handle(cfg, cat, local_rank, a.eps, a.upsample, a.path)
    blist = cfg["beam"] if isinstance(cfg["beam"], list) else [cfg["beam"]]
    outs = []
    for tsl, fsl in mine:
        outs.append(torch.empty(h.out_shape(tsl.stop - tsl.start, fsl.stop - fsl.start), dtype=cdt, device=dev))

    def step():
        for (tsl, fsl), o in zip(mine, outs):
            h.run_device(tsl.start, tsl.stop, fsl.start, fsl.stop, o.data_ptr())

    # First launches build per-geometry tables (bin order, twiddles, gather plans); then the W untimed
    # warm-up steps.  Small workloads (a C2 step is 1.5 ms) repeat identical untimed steps until 40 ms have
    # been queued, so that a small --warmup does not time the clock ramp instead of the kernels.
    t_settle = time.perf_counter()
    for _ in range(a.warmup):
        step()
    h.sync()
    extra = 0
    while a.light and time.perf_counter() - t_settle < 0.040 and extra < 64:
        step()
        h.sync()
        extra += 1
    h.reset_stats()
    # HIP events on the engine's own stream, attached to the spread dispatches themselves: on large grids
    # (a launch is hundreds of us) on EVERY launch (level 3); on C2-sized grids on the launches of one
    # time step in 16 (level 1), because even attached events idle the queue for 5-8 us around a 50 us
    # launch.  The other kernel families are timed in one extra, untimed step afterwards (level 2).
    h.enable_timing(1 if a.light else 3)

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    h.sync()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0  # this rank's own work, before it waits for the others
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [own]
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, own)

    st, tm = h.stats(), h.timing()
    h.reset_stats()
    breakdown = not a.no_breakdown and rank == 0 and len(mine) > 0
    if breakdown:
        h.enable_timing(2)
        step()
        h.sync()
        tm_all, st_all = h.timing(), h.stats()
    else:  # profiling runs: only launches shaped like the timed region's
        tm_all, st_all = None, st
    h.enable_timing(0)
    finite = all(bool(torch.isfinite(torch.view_as_real(o)).all().item()) for o in outs)
    vis_per_step = nbls * nfreq * ntimes  # the whole observation, all ranks together
    if a.as_rank is not None:
        vis_per_step = nbls * sum((t.stop - t.start) * (f.stop - f.start) for t, f in mine)
    value = vis_per_step * a.steps / elapsed

    if rank == 0:
        # ---- roofline of the spread kernel (the kernel BASELINE.json's metric names) ----------
        launches = max(st["spread_launches"], 1.0)
        timed = max(tm.get("spread_launches_timed", 0.0), 1.0)
        R8 = RB
        # transforms are 2-D for coplanar arrays AND for nearly flat ones, which run as K 2-D transforms per slice
        # (height terms, fv_sim.h wt_K); 3-D otherwise
        hterms = int(st.get("height_terms", 0))
        d = 2 if coplanar or hterms else 3
        my_times = sum(t.stop - t.start for t, _ in mine)
        # algorithmic bytes:  M (d R + T 2R)  +  T G1 2R   summed over launches
        spread_bytes = st["source_visits"] * 2 * R8 + (st["sources_above_horizon"] / max(my_times * a.steps, 1)) \
            * d * R8 * launches + st["spread_cells"] * 2 * R8
        spread_s = tm["spread"] * 1e-3 * launches / timed  # all launches (level 3: timed == launches)
        spread_kernel = "k_spread2d" if d == 2 else "k_spread3d"
        if d == 2 and os.environ.get("FFTVIS_HIP_SPREAD_CELL", "") != "1":
            # Nufft3::launch_spread picks the channel-group lane mapping once the catalog has >= 3
            # sources per 8 x 8-cell block (all launches of these workloads are chunks of >= 8 transforms)
            nax, nay = int(st["n2z"]) // 65536, int(st["n2z"]) % 65536
            if os.environ.get("FFTVIS_HIP_SPREAD_CELL") == "0" or nsrc >= 3 * ((nax + 7) // 8) * ((nay + 7) // 8):
                spread_kernel = "k_spread2d_cg"
                # ... and in fp64 the same walk with the accumulation on the matrix pipe (k_spread2d_mm)
                mm = os.environ.get("FFTVIS_HIP_SPREAD_MM")
                if R8 == 8 and (mm != "0" if mm is not None else "FFTVIS_HIP_SPREAD_CELL" not in os.environ):
                    spread_kernel = "k_spread2d_mm"
        if a.path == "type1" and breakdown:
            # lattice path: every (source, channel) pair is an entry with its own origin and 2 w
            # weights; timed in the extra step (event records around the launch)
            spread_kernel = "k_t1_spread"
            tpol = 4 if pol else 1
            entries = st_all["source_visits"] / tpol
            spread_bytes = st_all["source_visits"] * 2 * R8 + entries * (8 + 2 * st_all["w"] * R8) \
                + st_all["spread_cells"] * 2 * R8
            launches = max(st_all["spread_launches"], 1.0)
            spread_s = tm_all["spread"] * 1e-3
            tm = dict(tm, spread=tm_all["spread"])
            timed = launches
        ach = spread_bytes / spread_s / 1e9 if spread_s > 0 else 0.0
        # HBM traffic of the spread kernel from the committed PMC passes over this same command (rocprofv3
        # counters cannot be read from inside the process): only for the workload those passes ran -- and only if
        # the file still describes THIS run: same kernel family, same launches per time step, bytes that can belong
        # to the launches timed here.  A file that no longer matches fails the run instead of going quietly to null.
        traffic, traffic_src = None, None
        default_shape = not (a.nsrc or a.nfreq or a.ntimes) and a.upsample == 2.0 and dist is None and a.path == "type3" \
            and "FFTVIS_HIP_NO_HERMITIAN" not in os.environ and a.as_rank is None and a.array is None and a.z_scatter == 0
        pmf = pmc_file()
        if a.workload in ("C2", "C3") and default_shape and pmf and not os.environ.get("FFTVIS_BENCH_NO_PMC_CHECK"):
            pm = json.load(open(os.path.join(ROOT, pmf)))
            ent = pm["counters"].get(a.workload.lower(), {})
            k = ent.get(spread_kernel)
            per_time = launches / max(my_times * a.steps, 1)  # logical launches per time step of this run
            problems = []
            if k is None:
                problems.append(f"no entry for kernel family {spread_kernel} (file has {sorted(x for x in ent if x.startswith('k_'))})")
            else:
                nlog = ent.get("logical_launches", 0)
                if per_time <= 0 or abs(nlog / per_time - round(nlog / per_time)) > 1e-9 or nlog < per_time:
                    problems.append(f"{nlog} logical launches in the file are not whole time steps of this run's {per_time:g} launches per time step")
                traffic = (2 * k["FETCH_SIZE_KB_avg_per_launch"] + k["WRITE_SIZE_KB_avg_per_launch"]) * 1024
                alg = spread_bytes / launches
                if not 0.9 <= traffic / alg <= 1.6:
                    problems.append(f"traffic {traffic:.4g} B per launch against {alg:.4g} algorithmic bytes ({traffic / alg:.2f}x): not this kernel's launches")
            if problems:
                raise SystemExit(f"bench.py: {pmf} does not describe this run -- " + "; ".join(problems) +
                                 " -- re-collect it (tools/collect_profiles.sh, tools/refresh_profiles.py) or set FFTVIS_BENCH_NO_PMC_CHECK=1")
            traffic_src = (f"{pmf} (git blob {git_blob_id(os.path.join(ROOT, pmf))[:12]}; rocprofv3 --pmc, separate FETCH_SIZE / "
                           "WRITE_SIZE passes; 2*FETCH_SIZE+WRITE_SIZE per the gfx950 note; checked against this run's kernel family, "
                           "launches per time step and algorithmic bytes)")
        kern, fft, interp_rf = None, None, None
        if breakdown:
            l2 = max(st_all["spread_launches"], 1.0)
            fft_bytes = st_all["fft_cells"] * 2 * R8
            fam = ("spread", "fft", "interp", "strengths", "prep")
            total = max(sum(tm_all[k] for k in fam), 1e-9)
            kern = {
                "note": "per-family times from one extra single-stream step with event records around every launch",
                "spread_ms_per_launch": tm_all["spread"] / l2,
                "fft_ms_per_launch": tm_all["fft"] / l2,
                "interp_ms_per_launch": tm_all["interp"] / l2,
                "strengths_ms_per_launch": tm_all["strengths"] / l2,
                "prep_ms_per_step": tm_all["prep"],
                "launches_per_step": l2,
                "share_of_step": {k: tm_all[k] / total for k in fam},
                "grid_top_channel": {"n2": [int(st["n2x"]), int(st["n2y"])] + ([int(st["n2_3"])] if d == 3 else []),
                                     "active": [int(st["n2z"]) // 65536, int(st["n2z"]) % 65536] + ([int(st["na_3"])] if d == 3 else [])},
                "kernel_width": int(st["w"]),
                "height_terms": hterms,
                "height_terms_light_from": [int(st.get("height_terms_light_from", 0)), int(st.get("height_terms_lighter_from", 0))],
            }
            if tm_all["fft"] > 0:
                gbps = fft_bytes / (tm_all["fft"] * 1e-3) / 1e9
                # the pruned row FFT is the largest share of the step; same accounting: algorithmic
                # bytes of its passes over its summed pass durations (HIP events)
                fft = {
                    "kernel": "k_rowfft_st (x-pass + y-pass of the pruned 2-D FFT)" if d == 2 else
                              "k_rowfft_st / k_rowfft_dif (x-, y- and z-pass of the pruned 3-D FFT)",
                    "bound": "hbm",
                    "achieved": gbps,
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": gbps / HBM_PEAK_GBS,
                    "share_of_step": tm_all["fft"] / total,
                    "algorithmic_bytes_per_step": fft_bytes,
                }
                # the same passes against the vector pipe (fp64 MFMA has the same peak on gfx950 and a dense DFT costs
                # an order of magnitude more flops): flops priced as plain transforms, 5 n2 log2 n2 per line transformed
                fl = st_all.get("fft_flops", 0.0)
                if fl > 0:
                    peak_tf = 78.6 if R8 == 8 else 157.3
                    tfs = fl / (tm_all["fft"] * 1e-3) / 1e12
                    fft["flops"] = {"bound": "valu", "achieved": tfs, "peak": peak_tf, "unit": "TFLOP/s", "frac": tfs / peak_tf,
                                    "flops_per_step": fl,
                                    "note": "vector fp%d peak of MI355X_MICROARCH.md; since the column plan the passes move a third "
                                            "of the bytes they did and are bound by their instruction stream, not by HBM" % (8 * R8)}
            if tm_all["interp"] > 0 and a.path == "type3":
                # gather at the targets: algorithmic bytes = w^d grid values per (target, transform[, mirror side])
                # footprint + one output value per target and transform
                gb = st_all["interp_items"] * (int(st["w"]) ** d) * 2 * R8  # w^d footprint values per item
                gbps = gb / (tm_all["interp"] * 1e-3) / 1e9
                interp_rf = {
                    "kernel": "k_interp (gather of the transform grid at the baselines)",
                    "bound": "hbm",
                    "achieved": gbps,
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": gbps / HBM_PEAK_GBS,
                    "share_of_step": tm_all["interp"] / total,
                    "algorithmic_bytes_per_step": gb,
                    "note": "footprint rows are w-element pieces of 128-B lines: the lines fetched are about twice these bytes",
                }
        own_ms = [1e3 * t / a.steps for t in per_rank]
        h_closed = False
        res = {
            "metric": "simulated visibilities/sec (baselines x freqs x times) at eps="
                      + ("6e-8" if a.eps == 6e-8 else f"{a.eps:g}"),
            "value": value,
            "unit": "visibilities/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "settle_steps": extra,  # untimed repeats beyond --warmup (small workloads only, see above)
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if precision == 1 else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{a.workload}: {('scattered350 (no lattice, no repeated baseline vectors)' if a.array == 'scattered' else synth.CONFIGS[a.base][0])}"
                            f"{(' + %g m height scatter (non-coplanar: %s)' % (a.z_scatter, ('%d height terms, 2-D transforms' % hterms) if hterms else '3-D transform')) if a.z_scatter else ''}, {nsrc} sources, "
                            f"{nfreq} freqs, {ntimes} times, {nbls} baselines, "
                            f"{('%d basis beams (eigenbeam path), polarized' % len(blist)) if 'beam_coefs' in cfg else 'polarized table beam' if pol else 'unpolarized Airy beam'}, "
                            f"{a.path} NUFFT eps={a.eps:g} upsampfac={st.get('upsample_used', a.upsample):g}"
                            + (" (chosen by the engine)" if a.upsample == 0 else ""),
                "slices_per_step": nfreq * ntimes,
                "sharding": (f"ONLY the block of rank {a.as_rank} of {a.of_ranks}: " if a.as_rank is not None else "")
                            + "one observation, (time x freq) blocks per rank: "
                            + "; ".join(f"r{r}: t[{b[0][0].start}:{b[0][0].stop}) f[{b[0][1].start}:{b[0][1].stop})" if b else f"r{r}: -"
                                        for r, b in enumerate(blocks)),
                "per_rank_ms_per_step": own_ms,
                "imbalance_max_over_mean": max(own_ms) / (sum(own_ms) / len(own_ms)),
                "finite_output": finite,
            },
            "roofline": {
                "kernel": spread_kernel,
                "bound": "hbm",
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": spread_bytes / launches,
                "avg_launch_ms": tm["spread"] / timed,
                "launches_timed": timed,
                "launches_in_timed_region": launches,
                "lanes": int(st.get("lanes", 1)),
                "note": ("timed region: consecutive time steps run on two streams of equal priority, so this kernel shares the "
                         "GPU with the other time step's row passes / gather while it runs -- its duration here is that of a "
                         "kernel with part of the chip; `roofline_single_stream` is the same kernel with the chip to itself")
                        if int(st.get("lanes", 1)) > 1 and int(st.get("lane_mode", 1)) == 0 else None,
            },
            "roofline_fft": fft,
            "roofline_interp": interp_rf,
            "kernels": kern,
        }
        if breakdown and tm_all["spread"] > 0 and a.path == "type3":
            # the spread kernel with the chip to itself (the extra single-stream step): same launches, same bytes
            l2 = max(st_all["spread_launches"], 1.0)
            g1 = (spread_bytes / launches) / (tm_all["spread"] / l2 * 1e-3) / 1e9
            res["roofline_single_stream"] = {"kernel": spread_kernel, "bound": "hbm", "achieved": g1, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": g1 / HBM_PEAK_GBS, "avg_launch_ms": tm_all["spread"] / l2,
                                             "traffic": traffic,
                                             "what": "one extra step on ONE stream with event records around every launch (the step "
                                                     "`kernels`, `roofline_fft` and `roofline_interp` come from)"}
        if not a.no_e2e and dist is None and a.as_rank is None:
            # ---- host to host: what a caller of simulate_vis() waits for (never `value`) -----------------
            # numpy in, numpy out (reference wrapper.py:85-336 -> cpu_simulate.py:843-854 returns a host array):
            # catalog / beam / baseline upload, per-geometry tables, the step itself, and the visibilities' way
            # back to the host -- pinned in place and overlapped with the kernels (fv_sim.h drain_to_host).
            import fftvis_amd

            h.close()
            outs.clear()
            del cat
            torch.cuda.empty_cache()
            h_closed = True
            kw = dict(cfg, upsample_factor=a.upsample if a.upsample else "auto",
                      force_use_type3=a.path == "type3")
            walls, nbytes = [], 0
            for _ in range(2):
                t_e = time.perf_counter()
                v = fftvis_amd.simulate_vis(**kw)
                walls.append(time.perf_counter() - t_e)
                nbytes = v.nbytes
                del v
            res["e2e"] = {
                "what": "fftvis_amd.simulate_vis(**cfg): host arrays in, host array out, one call = one step",
                "first_call_s": walls[0],
                "second_call_s": walls[1],
                "ratio_to_device_resident_step": walls[1] / (elapsed / a.steps),
                "value": vis_per_step / walls[1],
                "unit": "visibilities/s",
                "output_bytes": nbytes,
                "d2h": "off" if os.environ.get("FFTVIS_HIP_D2H_OVERLAP") == "0" else
                       "caller's array pinned in place, one async copy per (channel, finished time step) on a copy stream",
            }
        if a.workload == "C3" and default_shape and not a.no_extras:
            # ---- side runs that ride in the driver's line so that they cannot rot unseen -----------------------
            if not h_closed:
                h.close()
                outs.clear()
                del cat
                torch.cuda.empty_cache()
                h_closed = True
            # (1) the generic type-3 engine: the same workload on an array WITHOUT a lattice or repeated baseline
            # vectors -- no column plan, no redundant-baseline gather (what finufft's contract promises nothing about)
            g = mini_run(synth.make_config("C3", array="scattered350"), local_rank, steps=2, warmup=1, breakdown=True)
            res["generic_array"] = dict(g, what="C3 on scattered350 (350 antennas uniform in HERA-350's footprint, all 61075 "
                                                "baselines distinct): type-3 NUFFT without the regular-array prunings",
                                        ratio_to_lattice_step=g["ms_per_step"] / res["ms_per_step"])
            # (2) the path the reference takes BY DEFAULT on this lattice array (type 1, cpu_simulate.py:634-681)
            t1 = mini_run(dict(synth.make_config("C3"), force_use_type3=False), local_rank, path="type1", steps=3, warmup=1)
            res["default_path_gpu"] = dict(t1, what="C3 through the lattice (type-1) path simulate_vis takes on this array "
                                                    "unless force_use_type3 (BASELINE.md 3.3)")
            # (3) C2 (configs[1]) as a 50-step mini-run: the small-grid regime, tracked round to round
            c2 = mini_run(synth.make_config("C2"), local_rank, steps=50, warmup=20, settle_s=0.04)
            res["c2_ms_per_step"] = c2["ms_per_step"]
            res["c2"] = c2
        if not a.no_cpu_baseline and dist is None:
            res["cpu_baseline"] = cpu_baseline(cfg, a.cpu_seconds)
    # ---- N > 1, host to host: what a caller of the sharded simulate_vis waits for (never `value`) -----------------
    # every rank: catalog broadcast from rank 0 into device memory, its block through the engine, delivered straight
    # into its slice of ONE shared-memory result while later time steps compute (parallel.simulate_vis_sharded)
    if dist is not None and not a.no_e2e and a.as_rank is None:
        # (a safety net around a part no 8-GPU node has run yet: if the host-to-host calls have not finished after
        # FFTVIS_BENCH_E2E_TIMEOUT seconds -- a collective stuck on some rank -- every rank leaves, rank 0 after printing
        # the line it already has, with the timeout recorded instead of `e2e_sharded`)
        import threading

        def give_up(why="timed out: the sharded host-to-host calls did not finish"):
            if rank == 0:
                res["e2e_sharded"] = {"error": why}
                print(json.dumps(res), flush=True)
            os._exit(0)

        watchdog = threading.Timer(float(os.environ.get("FFTVIS_BENCH_E2E_TIMEOUT", "240")), give_up)
        watchdog.daemon = True
        watchdog.start()
        h.close()
        outs.clear()
        del cat
        torch.cuda.empty_cache()
        kw = dict(cfg, upsample_factor=a.upsample if a.upsample else "auto", force_use_type3=a.path == "type3")
        if rank != 0:
            kw["ra"] = kw["dec"] = kw["fluxes"] = None
        walls = []
        try:
            for _ in range(2):
                dist.barrier()
                torch.cuda.synchronize()
                t_e = time.perf_counter()
                v = parallel.simulate_vis_sharded(device=local_rank, gather_to=0, via_host=backend != "nccl", **kw)
                dist.barrier()
                walls.append(time.perf_counter() - t_e)
                nbytes = v.nbytes if v is not None else 0
                fin = bool(np.isfinite(v[::max(1, v.shape[0] // 4)]).all()) if v is not None else True
                del v
        except Exception as e:  # (a rank that failed, or a peer that left: the timed line is still printed)
            give_up("failed on rank %d: %r" % (rank, e))
        watchdog.cancel()
        if rank == 0:
            res["e2e_sharded"] = {
                "what": "parallel.simulate_vis_sharded(**cfg) on all ranks: host arrays in on rank 0 (catalog broadcast into "
                        "every rank's HBM), every rank's block delivered straight into its slice of one shared-memory "
                        "result on the host; barrier to barrier, one call = one step",
                "first_call_s": walls[0],
                "second_call_s": walls[1],
                "ratio_to_device_resident_step": walls[1] / (elapsed / a.steps),
                "value": vis_per_step / walls[1],
                "unit": "visibilities/s",
                "output_bytes": nbytes,
                "finite_output": fin,
            }
    if rank == 0:
        if dist is not None and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, a.cpu_seconds)  # rank 0's host cores, as at N = 1 (the others wait)
        print(json.dumps(res), flush=True)
    h.close()  # (idempotent)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
