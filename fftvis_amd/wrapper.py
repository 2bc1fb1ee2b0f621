"""Public entry point: ``simulate_vis(..., backend="gpu")``.

Counterpart of the reference's src/fftvis/wrapper.py:16-336 for the one backend this package
provides.  Argument handling follows the reference line by line where the dependencies allow;
pyuvdata-specific steps (``UVBeam.interp``, ``prepare_beam_unpolarized``) are replaced by their
equivalents on this package's beam containers (core/beams.py ``describe_beam``).
"""

from __future__ import annotations

import numpy as np

from .core.simulate import SimulationEngine, default_accuracy_dict
from .core.beams import feed_index
from .core.utils import get_desired_chunks, validate_beam_idx


def create_beam_evaluator(backend: str = "gpu", **kwargs):
    """reference wrapper.py:16-48 -- here "gpu" is the implemented backend."""
    if backend == "gpu":
        from .gpu.beams import GPUBeamEvaluator

        ev = GPUBeamEvaluator(**kwargs)
        ev.beam_list = []
        ev.beam_idx = None
        return ev
    if backend == "cpu":
        raise NotImplementedError("fftvis_amd provides the gpu backend only; use fftvis for cpu")
    raise ValueError(f"Unsupported backend: {backend}")


def create_simulation_engine(backend: str = "gpu", **kwargs) -> SimulationEngine:
    """reference wrapper.py:51-82."""
    if backend == "gpu":
        from .gpu.gpu_simulate import GPUSimulationEngine

        return GPUSimulationEngine(**kwargs)
    if backend == "cpu":
        raise NotImplementedError("fftvis_amd provides the gpu backend only; use fftvis for cpu")
    raise ValueError(f"Unsupported backend: {backend}")


def device_memory_budget(device: int) -> int:
    """Bytes a run on ``device`` may plan with: what the device reports free PLUS what this process's own
    cached handle and NUFFT workspaces hold ON THAT DEVICE (``fv_device_bytes_on``) -- the next run reuses that
    handle's buffers or, if it needs another handle, releases them first (``_acquire_handle`` closes the idle
    handles of the device on a miss), so counting them as taken would shrink the budget after every large run
    (more source chunks, smaller time blocks, results differing at rounding level) although nothing else is
    using the memory.  Memory held on other devices is not this device's to give.  The reference measures
    available host RAM (wrapper.py:292-302), which has no such self-accounting."""
    import ctypes

    from . import _lib

    free, total, held = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(_lib.lib().fv_device_mem_info(int(device), ctypes.byref(free), ctypes.byref(total)))
    _lib.check(_lib.lib().fv_device_bytes_on(int(device), ctypes.byref(held)))
    return int(min(total.value, free.value + max(held.value, 0)))


def device_chunks(device, max_memory, min_chunks, beam_list, nax, nfeed, nant, nsrc, precision, source_buffer, nfreq):
    """``nchunks`` as the reference derives it (wrapper.py:292-302), against device memory."""
    nchunks, _ = get_desired_chunks(min(max_memory, device_memory_budget(device)), min_chunks, beam_list, nax, nfeed,
                                    nant, nsrc, precision, source_buffer=source_buffer, nfreq=nfreq)
    return nchunks


def simulate_vis(
    ants: dict,
    fluxes: np.ndarray,
    ra: np.ndarray,
    dec: np.ndarray,
    freqs: np.ndarray,
    times,
    beam,
    telescope_loc,
    beam_idx: np.ndarray = None,
    baselines: list = None,
    precision: int = 2,
    polarized: bool = False,
    eps: float = None,
    upsample_factor=2,
    beam_spline_opts: dict = None,
    use_feed: str = "x",
    flat_array_tol: float = 1e-6,
    interpolation_function: str = "az_za_map_coordinates",
    nprocesses: int | None = 1,
    nthreads: int | None = None,
    coord_method: str = "CoordinateRotationERFA",
    coord_method_params: dict | None = None,
    force_use_type3: bool = False,
    force_use_ray: bool = False,
    trace_mem: bool = False,
    backend: str = "gpu",
    max_memory=np.inf,
    min_chunks: int = 1,
    source_buffer=1.0,
    beam_coefs: np.ndarray = None,
    device: int = 0,
    coord_mgr=None,
    reference_compat: bool = True,
    astrom: np.ndarray = None,
    device_astrometry: bool = False,
) -> np.ndarray:
    """Visibilities (nfreqs, ntimes, nbls) or (nfreqs, ntimes, 2, 2, nbls); arguments as the
    reference's ``simulate_vis`` (wrapper.py:85-238).

    ``max_memory`` / ``min_chunks`` / ``source_buffer`` act as in the reference (wrapper.py:292-302),
    against DEVICE memory: the source axis is cut into at least ``min_chunks`` pieces, more if the
    per-time working set would not fit in min(max_memory, free device memory).  ``use_feed`` picks the
    feed of an E-field beam whose power an unpolarized run uses (wrapper.py:278-279).  ``coord_method``
    defaults to the reference's "CoordinateRotationERFA": the engine builds matvis' manager for it as the CPU
    engine does (cpu_simulate.py:686-709; matvis / astropy imported on first use) unless a ready one is passed
    as ``coord_mgr=``; ``reference_compat=False`` (extra) replaces the reference's forms for flipped two-beam
    baselines and the eigenbeam (l, k) term by the exact ones; ``astrom=`` / ``device_astrometry=True`` (extra) run
    the per-source astrometry on the device from per-time ERFA contexts; see ``GPUSimulationEngine.simulate``."""
    if eps is None:
        eps = default_accuracy_dict[precision]  # wrapper.py:241-242
    ants = {k: np.array(v) for k, v in ants.items()}
    beam_list = list(beam) if isinstance(beam, (list, tuple)) else [beam]
    beam_idx = validate_beam_idx(beam_idx, beam_coefs, len(beam_list), len(ants))
    if not polarized and beam_coefs is not None:  # wrapper.py:280-283
        raise ValueError(
            "Basis decomposition is not compatible with unpolarized simulations. Set polarized=True to use beam_coefs."
        )
    feed_index(use_feed)  # 'x' or 'y'
    nax = nfeed = 2 if polarized else 1
    engine = create_simulation_engine(backend=backend, device=device)
    nchunks = device_chunks(device, max_memory, min_chunks, beam_list, nax, nfeed, len(ants),
                            len(np.atleast_1d(ra)), precision, source_buffer, int(np.size(freqs)))
    return engine.simulate(
        ants=ants, freqs=np.asarray(freqs), fluxes=fluxes, beam_list=beam_list, beam_idx=beam_idx,
        ra=ra, dec=dec, times=times, telescope_loc=telescope_loc, baselines=baselines,
        precision=precision, polarized=polarized, eps=eps, upsample_factor=upsample_factor,
        beam_spline_opts=beam_spline_opts, flat_array_tol=flat_array_tol,
        interpolation_function=interpolation_function, nprocesses=nprocesses, nthreads=nthreads,
        coord_method=coord_method, coord_method_params=coord_method_params,
        force_use_type3=force_use_type3, force_use_ray=force_use_ray, trace_mem=trace_mem,
        nchunks=nchunks, source_buffer=source_buffer, beam_coefs=beam_coefs,
        coord_mgr=coord_mgr, use_feed=use_feed, reference_compat=reference_compat,
        astrom=astrom, device_astrometry=device_astrometry,
    )
