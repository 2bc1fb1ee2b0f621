"""Public entry point: ``simulate_vis(..., backend="gpu")``.

Counterpart of the reference's src/fftvis/wrapper.py:16-336 for the one backend this package
provides.  Argument handling follows the reference line by line where the dependencies allow;
pyuvdata-specific steps (``UVBeam.interp``, ``prepare_beam_unpolarized``) are replaced by their
equivalents on this package's beam containers (core/beams.py ``describe_beam``).
"""

from __future__ import annotations

import numpy as np

from .core.simulate import SimulationEngine, default_accuracy_dict
from .core.utils import validate_beam_idx


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
    coord_method: str = "SiderealRotation",
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
) -> np.ndarray:
    """Visibilities (nfreqs, ntimes, nbls) or (nfreqs, ntimes, 2, 2, nbls); arguments as the
    reference's ``simulate_vis`` (wrapper.py:85-238)."""
    if eps is None:
        eps = default_accuracy_dict[precision]  # wrapper.py:241-242
    ants = {k: np.array(v) for k, v in ants.items()}
    beam_list = list(beam) if isinstance(beam, (list, tuple)) else [beam]
    beam_idx = validate_beam_idx(beam_idx, beam_coefs, len(beam_list), len(ants))
    if not polarized and beam_coefs is not None:  # wrapper.py:280-283
        raise ValueError(
            "Basis decomposition is not compatible with unpolarized simulations. Set polarized=True to use beam_coefs."
        )
    engine = create_simulation_engine(backend=backend, device=device)
    return engine.simulate(
        ants=ants, freqs=np.asarray(freqs), fluxes=fluxes, beam_list=beam_list, beam_idx=beam_idx,
        ra=ra, dec=dec, times=times, telescope_loc=telescope_loc, baselines=baselines,
        precision=precision, polarized=polarized, eps=eps, upsample_factor=upsample_factor,
        beam_spline_opts=beam_spline_opts, flat_array_tol=flat_array_tol,
        interpolation_function=interpolation_function, nprocesses=nprocesses, nthreads=nthreads,
        coord_method=coord_method, coord_method_params=coord_method_params,
        force_use_type3=force_use_type3, force_use_ray=force_use_ray, trace_mem=trace_mem,
        nchunks=min_chunks, source_buffer=source_buffer, beam_coefs=beam_coefs,
        coord_mgr=coord_mgr,
    )
