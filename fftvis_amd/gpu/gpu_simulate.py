"""GPU simulation engine: the filled-in ``GPUSimulationEngine`` of the reference
(stub at src/fftvis/gpu/gpu_simulate.py:20-91).

``simulate`` has the *CPU engine's* signature and return layout
(src/fftvis/cpu/cpu_simulate.py:537-569, 850-854) because that is what ``simulate_vis`` calls
(src/fftvis/wrapper.py:308-336); the stub's own narrower signature could not accept that call.
Host work here is one-time setup in numpy; every per-(time, frequency) computation happens in
libfftvis_hip (hand-written HIP kernels, no library FFT) behind the C ABI of include/fftvis_hip.h.  There is no CPU
fallback: without the library or a GPU the calls raise.
"""

from __future__ import annotations

import ctypes
import atexit
import logging
import warnings

import numpy as np

from .. import _lib
from ..core import utils
from ..core.antenna_gridding import check_antpos_griddability
from ..core.beams import (airy_factors, checked_spline_order, describe_beam, feed_index, is_sampled_analytic,
                          table_tolerance)
from ..core.coords import SiderealRotation, eq_unit_vectors, julian_dates
from ..core.simulate import SimulationEngine, default_accuracy_dict

logger = logging.getLogger(__name__)


# Handles kept between ``simulate`` calls.  Creating and destroying a handle (streams, events, device
# buffers, per-geometry tables) costs ~15 ms, ten times the GPU work of a HERA-37 simulation -- and for a
# HERA-350 call ~85 ms of buffers, tables, launch lists and column plans that the next call on the same array
# would find ready; every ``fv_sim_set_*`` call fully replaces what it configures (re-setting what is already
# there keeps what was planned from it), so a handle whose creation parameters match can serve the next call.
# At most two idle handles, and only while this process holds less than FFTVIS_HIP_HANDLE_CACHE_BYTES of device
# memory (default: a quarter of the device's memory, at least 2 GiB; 0 = never keep one).
_IDLE_HANDLES: dict = {}
atexit.register(lambda: release_handles())


def _cache_limit(device: int = 0) -> int:
    import os

    env = os.environ.get("FFTVIS_HIP_HANDLE_CACHE_BYTES")
    if env is not None:
        return int(float(env))
    free, total = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(_lib.lib().fv_device_mem_info(int(device), ctypes.byref(free), ctypes.byref(total)))
    return max(2 * 1024**3, total.value // 4)


def release_handles():
    """Destroy the idle handles and this thread's cached stand-alone NUFFT workspace (and free their
    device memory)."""
    while _IDLE_HANDLES:
        _IDLE_HANDLES.popitem()[1].close()
    if _lib._lib is not None:  # only if the library was ever loaded
        _lib._lib.fv_release_workspaces()


def _acquire_handle(device, precision, eps, upsample_factor, polarized):
    key = (int(device), int(precision), float(eps), str(upsample_factor), bool(polarized))
    h = _IDLE_HANDLES.pop(key, None)
    # the other idle handles of this device hold memory that wrapper.device_memory_budget counts as the run's:
    # released now (a matching handle's buffers are reused instead), as is the stand-alone NUFFT workspace on a miss
    for k in [k for k in _IDLE_HANDLES if k[0] == key[0]]:
        _IDLE_HANDLES.pop(k).close()
    if h is None and _lib._lib is not None:
        _lib._lib.fv_release_workspaces()
    return key, (h if h is not None else SimHandle(device, precision, eps, upsample_factor, polarized))


def _return_handle(key, h):
    held = ctypes.c_int64(0)
    _lib.check(_lib.lib().fv_device_bytes_on(key[0], ctypes.byref(held)))
    limit = _cache_limit(key[0])
    if limit <= 0 or held.value > limit:
        h.close()
        return
    while len(_IDLE_HANDLES) >= 2:
        _IDLE_HANDLES.pop(next(iter(_IDLE_HANDLES))).close()
    _IDLE_HANDLES[key] = h


class SimHandle:
    """RAII wrapper of one ``fv_sim`` handle (one GPU context)."""

    def __init__(self, device: int, precision: int, eps: float, upsample_factor: float,
                 polarized: bool):
        self._L = _lib.lib()
        _lib.require_gpu()
        self.precision = precision
        self.eps = float(eps)
        self.polarized = bool(polarized)
        self.rdt = np.float32 if precision == 1 else np.float64
        self.cdt = np.complex64 if precision == 1 else np.complex128
        self._h = ctypes.c_void_p()
        if upsample_factor in (None, "auto"):
            upsample_factor = 0.0  # the engine picks 2 or 1.25 per run (include/fftvis_hip.h, fv_sim_create)
        _lib.check(self._L.fv_sim_create(ctypes.byref(self._h), device, precision, float(eps),
                                         float(upsample_factor), int(polarized)))
        self.nbls = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.fv_sim_destroy(self._h)
            self._h = None

    __del__ = close

    # -- uploads ---------------------------------------------------------------------------
    def set_sources(self, eq, flux, polarized_sky: bool):
        eq = np.ascontiguousarray(eq, dtype=self.rdt)
        flux = np.ascontiguousarray(flux, dtype=self.cdt if polarized_sky else self.rdt)
        _lib.check(self._L.fv_sim_set_sources(self._h, eq.shape[1], flux.shape[1], _lib.ptr(eq),
                                              _lib.ptr(flux), int(polarized_sky), 0))

    def set_sources_device(self, nsrc, nfreq, eq_ptr, flux_ptr, polarized_sky: bool):
        """Device pointers (e.g. torch tensors filled by an RCCL broadcast)."""
        _lib.check(self._L.fv_sim_set_sources(self._h, nsrc, nfreq, _lib.ptr(eq_ptr),
                                              _lib.ptr(flux_ptr), int(polarized_sky), 1))

    def set_times(self, rot):
        rot = np.ascontiguousarray(rot, dtype=np.float64)
        _lib.check(self._L.fv_sim_set_times(self._h, rot.shape[0], _lib.ptr(rot)))

    def set_astrom(self, astrom):
        """(ntimes, 31) float64 eraASTROM contexts: per-source astrometry runs on the device (fv_sim_set_astrom)."""
        a = np.ascontiguousarray(astrom, dtype=np.float64)
        _lib.check(self._L.fv_sim_set_astrom(self._h, a.shape[0], _lib.ptr(a)))

    def set_topo(self, topo):
        topo = np.ascontiguousarray(topo, dtype=self.rdt)
        _lib.check(self._L.fv_sim_set_topo(self._h, topo.shape[0], topo.shape[2], _lib.ptr(topo), 0))

    def set_freqs(self, freqs):
        f = np.ascontiguousarray(freqs, dtype=np.float64)
        _lib.check(self._L.fv_sim_set_freqs(self._h, f.size, _lib.ptr(f)))

    def set_array(self, rotation_matrix, bls, is_coplanar: bool):
        R = np.ascontiguousarray(rotation_matrix, dtype=np.float64)
        b = np.ascontiguousarray(bls, dtype=np.float64)
        self.nbls = b.shape[1]
        _lib.check(self._L.fv_sim_set_array(self._h, _lib.ptr(R), self.nbls, _lib.ptr(b),
                                            int(is_coplanar)))

    def set_array_type1(self, basis_matrix, bls_int, n_modes: int):
        B = np.ascontiguousarray(basis_matrix, dtype=np.float64)
        b = np.ascontiguousarray(bls_int[:2], dtype=np.int32)
        self.nbls = b.shape[1]
        _lib.check(self._L.fv_sim_set_array_type1(self._h, _lib.ptr(B), self.nbls, _lib.ptr(b),
                                                  int(n_modes)))

    def set_beams(self, beam_list, freqs, order: int = 1, use_feed: str = "x"):
        _lib.check(self._L.fv_sim_set_nbeams(self._h, len(beam_list)))
        if beam_list and all(is_sampled_analytic(b) for b in beam_list):
            order = 3  # only third-party analytic beams: their tables (if any) are ours to lay out -- cubic
        for i, beam in enumerate(beam_list):
            d = describe_beam(beam, self.polarized, np.asarray(freqs, dtype=float), use_feed, order,
                              table_tolerance(self.eps))
            if d[0] == "airy":  # this package's AiryBeam, or a third-party object PROVEN to be that form x factors
                fac = airy_factors(d)
                _lib.check(self._L.fv_sim_set_beam_airy_scaled(self._h, i, d[1], _lib.ptr(fac), float(fac[8])))
            else:
                tab = d[1]
                _lib.check(self._L.fv_sim_set_beam_table(self._h, i, tab.shape[0], tab.shape[-2],
                                                         tab.shape[-1], float(d[2]), _lib.ptr(tab),
                                                         int(order)))

    def set_beam_pairs(self, pairs, pair_idx, pair_flip):
        bi = np.array([p[0] for p in pairs], dtype=np.int32)
        bj = np.array([p[1] for p in pairs], dtype=np.int32)
        off = np.zeros(len(pairs) + 1, dtype=np.int64)
        idx, flp = [], []
        for n, p in enumerate(pairs):
            ii = np.asarray(pair_idx[p], dtype=np.int32)
            off[n + 1] = off[n] + ii.size
            idx.append(ii)
            flp.append(np.asarray(pair_flip[p], dtype=np.int8))
        idx = np.ascontiguousarray(np.concatenate(idx) if idx else np.zeros(0, np.int32))
        flp = np.ascontiguousarray(np.concatenate(flp) if flp else np.zeros(0, np.int8))
        if idx.size == 0:  # keep the pointers valid
            idx, flp = np.zeros(1, np.int32), np.zeros(1, np.int8)
        _lib.check(self._L.fv_sim_set_beam_pairs(self._h, len(pairs), _lib.ptr(bi), _lib.ptr(bj),
                                                 _lib.ptr(off), _lib.ptr(idx), _lib.ptr(flp)))

    def set_reference_compat(self, on: bool = True):
        """The reference's forms for flipped two-beam baselines / the eigenbeam (l, k) term, or the exact ones
        (fv_sim_set_reference_compat)."""
        _lib.check(self._L.fv_sim_set_reference_compat(self._h, int(bool(on))))

    def set_chunking(self, nchunks: int = 1, source_buffer: float = 1.0):
        """Source chunks per time step and the above-horizon buffer fraction (fv_sim_set_chunking)."""
        _lib.check(self._L.fv_sim_set_chunking(self._h, int(nchunks), float(source_buffer)))

    def set_basis(self, beam_coefs, ant1_idxs, ant2_idxs):
        c = np.ascontiguousarray(beam_coefs, dtype=self.cdt)
        a1 = np.ascontiguousarray(ant1_idxs, dtype=np.int32)
        a2 = np.ascontiguousarray(ant2_idxs, dtype=np.int32)
        _lib.check(self._L.fv_sim_set_basis(self._h, c.shape[0], c.shape[1], c.shape[2],
                                            _lib.ptr(c), _lib.ptr(a1), _lib.ptr(a2)))

    # -- execution ----------------------------------------------------------------------------
    def out_shape(self, nt, nf):
        return (nf, nt, 2, 2, self.nbls) if self.polarized else (nf, nt, self.nbls)

    def run(self, t0, t1, f0, f1, out=None, shared=False):
        """Times [t0,t1) x freqs [f0,f1) -> host array in the reference's final layout.

        ``out``: deliver into this array instead of a new one -- a block of a larger result, e.g.
        ``vis[fsl, tsl]`` (reference cpu_simulate.py:846-847): shape ``out_shape``, this engine's dtype, every axis
        but the first (channels) contiguous (``fv_sim_run_into``).  ``shared``: other processes fill the rest of the
        underlying array (a result in shared memory, one block per rank)."""
        shape = self.out_shape(t1 - t0, f1 - f0)
        if out is None:
            out = np.empty(shape, dtype=self.cdt)
            _lib.check(self._L.fv_sim_run(self._h, t0, t1, f0, f1, _lib.ptr(out), 0))
            return out
        if out.shape != shape or out.dtype != self.cdt or not out.flags.writeable:
            raise ValueError(f"out must be a writeable {np.dtype(self.cdt).name} array of shape {shape}, got {out.dtype} {out.shape}")
        inner = np.empty(shape[1:], dtype=self.cdt).strides
        if out.size and (out.strides[1:] != inner or out.strides[0] % out.itemsize or out.strides[0] < inner[0] * shape[1]):
            raise ValueError("out: only the first (channel) axis may be strided; the rest must be C-contiguous")
        if out.size:
            _lib.check(self._L.fv_sim_run_into(self._h, t0, t1, f0, f1, ctypes.c_void_p(out.ctypes.data),
                                               out.strides[0] // out.itemsize, int(bool(shared))))
        return out

    def run_device(self, t0, t1, f0, f1, out_ptr):
        """Enqueue only; ``out_ptr`` is a device buffer of out_shape() complex elements."""
        _lib.check(self._L.fv_sim_run(self._h, t0, t1, f0, f1, _lib.ptr(out_ptr), 1))

    def sync(self):
        _lib.check(self._L.fv_sim_sync(self._h))

    def stats(self):
        v = np.zeros(20)
        _lib.check(self._L.fv_sim_stats(self._h, _lib.ptr(v), 20))
        # ("n2z" is historical: active cells na_x * 65536 + na_y; the third dimension's sizes are n2_3 / na_3)
        keys = ["spread_launches", "spread_cells", "source_visits", "fft_cells", "interp_items",
                "sources_above_horizon", "n2x", "n2y", "n2z", "w", "upsample_used", "max_above_horizon", "fft_flops",
                "n2_3", "na_3", "height_terms", "lanes", "lane_mode", "height_terms_light_from", "height_terms_lighter_from"]
        return dict(zip(keys, v))

    def reset_stats(self):
        _lib.check(self._L.fv_sim_reset_stats(self._h))

    def enable_timing(self, on=True):
        _lib.check(self._L.fv_sim_enable_timing(self._h, int(on)))

    def timing(self):
        v = np.zeros(6)
        _lib.check(self._L.fv_sim_timing(self._h, _lib.ptr(v), 6))
        return dict(zip(["spread", "fft", "interp", "strengths", "prep", "spread_launches_timed"], v))


def register_with_reference() -> bool:
    """Make ``isinstance(engine, fftvis.core.simulate.SimulationEngine)`` hold where the reference is
    installed (virtual-subclass registration on its ABC, core/simulate.py:22).  Lazy: importing this module
    registers only if ``fftvis.core.simulate`` is ALREADY imported (importing the reference pulls in finufft,
    matvis, astropy and pyuvdata -- seconds, for an ``ABC.register``), the first ``GPUSimulationEngine()`` tries
    again if the caller has imported ``fftvis`` since, and the stub module of INTEGRATION.md, route A, calls this
    function itself (which does import ``fftvis.core.simulate``).  Only a missing reference is swallowed: any other
    error of that import is the caller's to see."""
    global _REGISTERED
    import sys

    mod = sys.modules.get("fftvis.core.simulate")
    if mod is None:
        import importlib

        try:
            mod = importlib.import_module("fftvis.core.simulate")
        except ImportError:  # the reference (or one of its dependencies) is not installed
            return False
    mod.SimulationEngine.register(GPUSimulationEngine)
    _REGISTERED = True
    return True


_REGISTERED = False


def prepare_array(ants: dict, baselines, flat_array_tol: float, real_dtype):
    """Plane-fit rotation, rotated baselines in seconds, coplanarity flag
    (reference cpu_simulate.py:628-659)."""
    antkey_to_idx = {a: i for i, a in enumerate(ants)}
    antvecs = np.array([ants[a] for a in ants], dtype=real_dtype)
    R = np.ascontiguousarray(utils.get_plane_to_xy_rotation_matrix(antvecs).T)
    rot = R @ antvecs.T
    i0 = np.array([antkey_to_idx[b[0]] for b in baselines], dtype=int)
    i1 = np.array([antkey_to_idx[b[1]] for b in baselines], dtype=int)
    bls = (rot[:, i1] - rot[:, i0]).reshape(3, len(baselines)).astype(float)
    is_coplanar = bool(np.all(np.abs(bls[2]) <= flat_array_tol))
    bls = (bls / utils.speed_of_light).astype(real_dtype)
    return R.astype(real_dtype), bls, is_coplanar


class GPUSimulationEngine(SimulationEngine):
    """MI355X implementation of the simulation engine."""

    def __init__(self, device: int = 0):
        import sys

        if not _REGISTERED and "fftvis" in sys.modules:  # the caller uses the reference: be an instance of ITS ABC
            register_with_reference()
        self.device = device

    def simulate(
        self,
        ants: dict,
        freqs: np.ndarray,
        fluxes: np.ndarray,
        beam_list: list,
        ra: np.ndarray,
        dec: np.ndarray,
        times,
        telescope_loc,
        baselines: list = None,
        beam_idx: np.ndarray = None,
        precision: int = 2,
        polarized: bool = False,
        eps: float = None,
        upsample_factor=2,
        beam_spline_opts: dict = None,
        flat_array_tol: float = 1e-6,
        interpolation_function: str = "az_za_map_coordinates",
        nprocesses: int | None = 1,
        nthreads: int | None = None,
        coord_method: str = "CoordinateRotationERFA",
        coord_method_params: dict | None = None,
        force_use_ray: bool = False,
        force_use_type3: bool = False,
        trace_mem: bool = False,
        enable_memory_monitor: bool = False,
        nchunks: int = 1,
        source_buffer=1.0,
        beam_coefs: np.ndarray = None,
        coord_mgr=None,
        time_idx: slice = slice(None),
        freq_idx: slice = slice(None),
        use_feed: str = "x",
        catalog_device=None,
        reference_compat: bool = True,
        astrom: np.ndarray = None,
        device_astrometry: bool = False,
        out: np.ndarray = None,
        out_shared: bool = False,
    ) -> np.ndarray:
        """Simulate visibilities on the GPU.

        Same arguments and return value as ``CPUSimulationEngine.simulate``
        (reference cpu_simulate.py:537-854).  Differences, all documented in DESIGN.md:

        * like the reference, a flat array whose antennas sit on a lattice takes the type-1 path
          unless ``force_use_type3`` (cpu_simulate.py:634-637); eigenbeam runs always use type 3
          here;
        * ``coord_method``: as in the reference (cpu_simulate.py:686-709) the engine builds matvis' coordinate
          manager ``CoordinateRotation._methods[coord_method]`` itself when the caller passes none (matvis and
          astropy are imported lazily, here only; without them the call raises ValueError) and uses its per-time
          topocentric vectors (``rotate(ti)`` -> ``all_coords_topo``) verbatim, streamed to the device one time
          block at a time; a caller-built manager can be handed over as ``coord_mgr`` (extra).  The
          mean-sidereal rotation of ``core/coords.py`` (no precession / nutation / aberration: ~0.35 deg off
          ICRS positions at 2025 epochs) runs only when asked for by name, ``coord_method="SiderealRotation"``;
        * ``astrom`` / ``device_astrometry`` (extra; SURVEY section 8 f3 -- the coordinate manager on the device):
          ``astrom`` = (ntimes, 31) float64, one ERFA ``eraASTROM`` context per time (``erfa.apco13`` / astropy's
          ``erfa_astrom.apco`` fill it in microseconds); the device applies it to every source -- light deflection,
          aberration, bias-precession-nutation, Earth rotation, polar motion, diurnal aberration, horizon frame,
          refraction (``fv_sim_set_astrom``) -- so no (ntimes, 3, nsrc) vectors are computed or streamed on the host.
          ``device_astrometry=True`` builds the contexts here with astropy (the context astropy's own ICRS -> AltAz
          uses; raises ValueError without astropy) instead of building the matvis manager.  Default off: the
          reference's manager stays the default route.  Unpinned against ERFA in this pipeline;
        * ``nchunks`` splits the source axis exactly like the reference's chunk loop
          (cpu_simulate.py:939,1024,1069): every time step processes the catalog in ``nchunks`` pieces
          whose visibilities accumulate on the device; ``source_buffer`` sizes the above-horizon
          arrays of a piece and a piece that overflows them raises, as matvis does.  On top of that the
          engine walks the time axis in blocks when the output block would not fit in free device
          memory;
        * ``nprocesses/nthreads/force_use_ray/trace_mem/enable_memory_monitor`` are CPU scheduling /
          tracing knobs with no meaning on one GPU (``n_threads`` "not used in GPU implementation",
          reference gpu/nufft.py:38): accepted, unused;
        * ``beam_coefs`` (eigenbeams): like the reference, the (l, k) term reuses V_kl transposed
          (exact for real-valued basis beams, reference cpu_simulate.py:464-468);
        * ``reference_compat`` (extra, default True = the reference's arithmetic): False switches the two places
          where the reference departs from the exact symmetry of the visibilities to the
          exact forms -- flipped baselines of a two-beam polarized pair become V_ij(-b)^H (conjugated AND feed
          block transposed; the reference only conjugates, cpu_simulate.py:298), and the eigenbeam (l, k) term
          becomes conj(V_kl(-b))^T (one more gather at -b; exact for complex basis beams too);
        * ``beam_spline_opts``: orders 0 .. 5 as scipy.ndimage.map_coordinates takes them (1 = bilinear, also when
          None; 3 = cubic B-spline; ``kx/ky`` of ``az_za_simple`` are read the same way -- both interpolation
          functions of the reference are regular-grid splines of that order and map onto the same device
          interpolant); other orders raise ValueError;
        * ``use_feed`` (extra; the reference's wrapper applies it before the engine,
          wrapper.py:278-279): the feed whose power pattern an unpolarized run takes from an E-field beam;
        * ``upsample_factor``: 2 (default, as the reference) or 1.25 are used as given; ``None`` /
          ``"auto"`` (extra) lets the engine pick per run -- 1.25 when eps >= 1e-8 (fp32: 1e-4) and
          the fine grid is large (HERA-350 class arrays: ~2x faster), else 2;
        * ``time_idx`` / ``freq_idx`` (extra) restrict the run to a block, which is how ranks
          shard a simulation across GPUs; ``catalog_device`` (extra; ``parallel.DeviceCatalog``) is a
          catalog already resident on this GPU -- e.g. received by an RCCL broadcast -- used instead
          of ``ra / dec / fluxes`` (which may then be None);
        * ``out`` (extra): deliver the block into this array -- typically a view ``vis[freq_idx, time_idx]`` of the
          whole result, the reference's ``vis[tc][..., fc] = future`` (cpu_simulate.py:846-847) without the copy --
          of the block's shape and dtype, only its channel axis strided; it is pinned in place and filled while the
          run computes.  ``out_shared`` (extra): the underlying array is shared memory that other ranks fill too
          (``parallel.simulate_vis_sharded``).
        """
        beam_order = checked_spline_order(beam_spline_opts)
        if interpolation_function not in ("az_za_map_coordinates", "az_za_simple"):
            raise ValueError(f"unknown interpolation_function {interpolation_function!r}")
        feed_index(use_feed)  # 'x' or 'y', else ValueError
        nchunks = int(nchunks)
        if nchunks < 1:
            raise ValueError("nchunks must be >= 1")
        if not 0.0 < float(source_buffer) <= 1.0:
            raise ValueError("source_buffer must be in (0, 1]")
        if astrom is None and device_astrometry and coord_mgr is None and coord_method != "SiderealRotation":
            from ..core.coords import erfa_astrom_context

            astrom = erfa_astrom_context(times, telescope_loc)
        if astrom is not None:
            astrom = np.ascontiguousarray(astrom, dtype=np.float64)
            if astrom.ndim != 2 or astrom.shape[1] != 31 or astrom.shape[0] != len(julian_dates(times)):
                raise ValueError("astrom must have shape (ntimes, 31): one eraASTROM context per time")
            if coord_mgr is not None:
                raise ValueError("pass either astrom= (device astrometry) or coord_mgr=, not both")
        if coord_mgr is None and astrom is None and coord_method != "SiderealRotation" and catalog_device is not None:
            raise ValueError(
                "a device-resident catalog carries no ra / dec for a matvis coordinate manager: pass coord_mgr= "
                "or coord_method='SiderealRotation'")
        freqs = np.asarray(freqs)
        nfreqs, ntimes, nbeam, nant = np.size(freqs), len(julian_dates(times)), len(beam_list), len(ants)
        real_dtype = np.float32 if precision == 1 else np.float64
        complex_dtype = np.complex64 if precision == 1 else np.complex128
        if eps is None:
            eps = default_accuracy_dict[precision]
        if upsample_factor == 1.25 and eps < (1e-8 if precision == 2 else 1e-4):
            # finufft's upsampfac = 1.25 has the same floor (its kernel width is capped likewise)
            warnings.warn(
                f"upsample_factor=1.25 delivers about {1e-8 if precision == 2 else 1e-4:g} at best "
                f"(eps={eps:g} was asked for); use upsample_factor=2 or 'auto' for tighter tolerances.",
                RuntimeWarning, stacklevel=2)
        # precision = 1 rounds ra/dec/freqs to float32 first (reference cpu_simulate.py:601-606)
        if catalog_device is None:
            ra = np.asarray(ra).astype(real_dtype)
            dec = np.asarray(dec).astype(real_dtype)
            nsrc = int(ra.size)
        else:
            nsrc = int(catalog_device.nsrc)
        freqs = freqs.astype(real_dtype)
        ants = {k: np.asarray(v) for k, v in ants.items()}

        beam_idx = utils.validate_beam_idx(beam_idx, beam_coefs, nbeam, nant)
        if baselines is None:
            baselines = [red[0] for red in utils.get_pos_reds(ants, include_autos=True)]
        if catalog_device is None:
            coherency, polarized_sky = utils.prepare_source_catalog(np.asarray(fluxes), polarized)
            if coherency.shape[0] != nsrc or coherency.shape[1] != nfreqs:
                raise ValueError("fluxes must have shape (nsources, nfreqs[, 4])")
        else:
            _check_device_catalog(catalog_device, nfreqs, polarized, precision, self.device)

        if coord_mgr is None and astrom is None and coord_method != "SiderealRotation":
            # the reference's own call (wrapper.py:308-336 passes no manager): build matvis' manager exactly
            # as the CPU engine does (cpu_simulate.py:686-709) and stream its vectors like a caller's
            coord_mgr = build_coord_mgr(coord_method, coord_method_params, coherency.astype(complex_dtype, copy=False),
                                        times, telescope_loc, ra, dec, precision, source_buffer, nchunks)

        # lattice arrays -> type 1 (reference cpu_simulate.py:634-637, 661-681)
        antvecs = np.array([ants[a] for a in ants], dtype=real_dtype)
        is_gridded = False
        if not force_use_type3 and beam_coefs is None and np.abs(antvecs[:, -1]).max() <= flat_array_tol:
            is_gridded, gridded_antpos, basis_matrix = check_antpos_griddability(ants)
        if is_gridded:
            bls_int = np.round(np.array(
                [gridded_antpos[bl[1]] - gridded_antpos[bl[0]] for bl in baselines]).T).astype(int)
            bls_int = bls_int.reshape(3, len(baselines))
            n_modes = 2 * int(np.round(np.max(np.abs(bls_int)))) + 1 if len(baselines) else 1
            basis_matrix = (basis_matrix / utils.speed_of_light).astype(real_dtype)
        else:
            R, bls, is_coplanar = prepare_array(ants, baselines, flat_array_tol, real_dtype)
        antnums = list(ants.keys())
        use_basis = beam_coefs is not None
        if use_basis:
            if not polarized:  # reference wrapper.py:280-283
                raise ValueError(
                    "Basis decomposition is not compatible with unpolarized simulations. Set polarized=True to use beam_coefs."
                )
            beam_coefs = np.asarray(beam_coefs)
            if beam_coefs.shape != (nant, nbeam, nfreqs):
                raise ValueError("beam_coefs must have shape (nant, nbasis, nfreqs)")
            # per-baseline antenna rows of beam_coefs (reference cpu_simulate.py:920-921)
            ant1_idxs = np.array([antnums.index(bl[0]) for bl in baselines])
            ant2_idxs = np.array([antnums.index(bl[1]) for bl in baselines])
        else:
            pairs, pair_idx, pair_flip = utils.prepare_beam_evaluation(antnums, baselines, beam_idx)

        key, h = _acquire_handle(self.device, precision, eps, upsample_factor, polarized)
        ok = False
        try:
            if catalog_device is None:
                h.set_sources(eq_unit_vectors(ra.astype(float), dec.astype(float)), coherency, polarized_sky)
            else:
                h.set_sources_device(nsrc, nfreqs, catalog_device.eq.data_ptr(), catalog_device.flux.data_ptr(),
                                     catalog_device.polarized_sky)
            if astrom is not None:
                h.set_astrom(astrom)
            elif coord_mgr is None:
                h.set_times(SiderealRotation(times, telescope_loc).matrices())
            h.set_freqs(freqs.astype(float))
            if is_gridded:
                logger.info("Using gridded coordinates for the array. Type 1 transform will be used.")
                h.set_array_type1(basis_matrix.astype(float), bls_int, n_modes)
            else:
                h.set_array(R.astype(float), bls.astype(float), is_coplanar)
            h.set_beams(beam_list, freqs.astype(float), beam_order, use_feed)
            h.set_chunking(min(nchunks, max(nsrc, 1)), float(source_buffer))
            h.set_reference_compat(reference_compat)
            if use_basis:
                h.set_basis(beam_coefs, ant1_idxs, ant2_idxs)
            else:
                h.set_beam_pairs(pairs, pair_idx, pair_flip)
            t0, t1, _ = time_idx.indices(ntimes)
            f0, f1, _ = freq_idx.indices(nfreqs)
            # Walk the time axis in blocks whose output fits comfortably in free device memory (the
            # reference holds the whole (nt, nbls, nfeeds, nfeeds, nf) block in host RAM,
            # cpu_simulate.py:909-911); a coordinate manager's vectors are streamed block by block
            # instead of being stacked for all times on the host.
            nblk_t = _time_block(self.device, t1 - t0, f1 - f0, len(baselines), polarized, precision,
                                 nsrc if coord_mgr is not None else 0)
            if coord_mgr is not None:
                coord_mgr.setup()
            if out is not None and (out.shape != h.out_shape(t1 - t0, f1 - f0) or out.dtype != complex_dtype):
                raise ValueError(f"out must be a {np.dtype(complex_dtype).name} array of shape "
                                 f"{h.out_shape(t1 - t0, f1 - f0)}, got {out.dtype} {out.shape}")
            if nblk_t >= t1 - t0 and coord_mgr is None:
                vis = h.run(t0, t1, f0, f1, out=out, shared=out_shared)
            else:
                # every time block lands in its slice of the result (fv_sim_run_into): no block-sized temporary, no
                # second host copy
                vis = out if out is not None else np.empty(h.out_shape(t1 - t0, f1 - f0), dtype=complex_dtype)
                for tb in range(t0, t1, max(nblk_t, 1)):
                    te = min(t1, tb + max(nblk_t, 1))
                    if coord_mgr is not None:
                        h.set_topo(_topo_from_coord_mgr(coord_mgr, range(tb, te)))
                        h.run(0, te - tb, f0, f1, out=vis[:, tb - t0:te - t0], shared=out_shared)
                    else:
                        h.run(tb, te, f0, f1, out=vis[:, tb - t0:te - t0], shared=out_shared)
            ok = True
        finally:
            if ok:
                _return_handle(key, h)
            else:  # an error may have left it half configured
                h.close()
        return vis if out is not None else vis.astype(complex_dtype, copy=False)

    def _evaluate_vis_chunk(self, time_idx: slice, freq_idx: slice, **kw) -> np.ndarray:
        """One (time x freq) block in the reference's scratch layout
        (nt_here, nbls, nfeeds, nfeeds, nf_here) (reference cpu_simulate.py:909-911,1071).
        Takes the keyword arguments of ``simulate`` plus the block."""
        polarized = kw.get("polarized", False)
        final = self.simulate(time_idx=time_idx, freq_idx=freq_idx, **kw)
        if polarized:  # inverse of np.transpose(vis, (4, 0, 2, 3, 1)), cpu_simulate.py:851
            return np.transpose(final, (1, 4, 2, 3, 0))
        return np.moveaxis(final, 0, 2)[:, :, None, None, :]


def build_coord_mgr(coord_method, coord_method_params, coherency, times, telescope_loc, ra, dec, precision,
                    source_buffer, nchunks):
    """The matvis coordinate manager the CPU engine builds at reference cpu_simulate.py:686-709, built the same
    way: ``CoordinateRotation._methods[coord_method](flux=, times=, telescope_loc=, skycoords=, precision=,
    source_buffer=, chunk_size=, **coord_method_params)``, BCRS fixed up front when ``update_bcrs_every``
    exceeds the observation's span.  matvis / astropy are imported here, on first use, because they are the
    reference's dependencies, not this backend's; without them the request is refused, never approximated."""
    try:
        from astropy import units as un
        from astropy.coordinates import SkyCoord
        from astropy.time import Time
        from matvis.core.coords import CoordinateRotation
    except ImportError as e:
        raise ValueError(
            f"coord_method={coord_method!r} needs matvis / astropy ({e}): install them (they are dependencies of "
            "fftvis), pass a matvis coordinate manager as coord_mgr= (its per-time topocentric vectors are used "
            "verbatim), or ask for the mean-sidereal approximation by name with coord_method='SiderealRotation' "
            "(no precession / nutation / aberration)") from e
    if coord_method not in CoordinateRotation._methods:
        raise ValueError(f"unknown coord_method {coord_method!r}")
    if isinstance(times, np.ndarray):  # cpu_simulate.py:686-687
        times = Time(times, format="jd")
    chunk_size = int(np.ceil(dec.size / nchunks))  # :689
    mgr = CoordinateRotation._methods[coord_method](
        flux=coherency,
        times=times,
        telescope_loc=telescope_loc,
        skycoords=SkyCoord(ra=ra * un.rad, dec=dec * un.rad, frame="icrs"),
        precision=precision,
        source_buffer=source_buffer,
        chunk_size=chunk_size,
        **(coord_method_params or {}),
    )
    if getattr(mgr, "update_bcrs_every", 0) > (times[-1] - times[0]).to(un.s):  # :706-709
        mgr._set_bcrs(0)
    return mgr


def _check_device_catalog(cat, nfreqs, polarized, precision, device):
    """Raw device pointers go to ``fv_sim_set_sources(on_device=1)``: everything the C side cannot see is
    checked here -- dtypes of the run's precision, contiguity, shapes, and the device the tensors live on."""
    import torch

    rdt = torch.float32 if precision == 1 else torch.float64
    cdt = torch.complex64 if precision == 1 else torch.complex128
    if cat.nfreq != nfreqs or (cat.polarized_sky and not polarized):
        raise ValueError("catalog_device does not match freqs / polarized")
    want_flux = (cat.nsrc, nfreqs, 2, 2) if cat.polarized_sky else (cat.nsrc, nfreqs)
    if tuple(cat.eq.shape) != (3, cat.nsrc) or tuple(cat.flux.shape) != want_flux:
        raise ValueError(f"catalog_device: eq must be (3, nsrc) and flux {want_flux}, got {tuple(cat.eq.shape)} / "
                         f"{tuple(cat.flux.shape)}")
    if cat.eq.dtype != rdt or cat.flux.dtype != (cdt if cat.polarized_sky else rdt):
        raise ValueError(f"catalog_device was built for another precision: eq {cat.eq.dtype}, flux {cat.flux.dtype}, "
                         f"run precision={precision}")
    if not (cat.eq.is_contiguous() and cat.flux.is_contiguous()):
        raise ValueError("catalog_device tensors must be contiguous")
    for t in (cat.eq, cat.flux):
        if t.device.type != "cuda" or (t.device.index or 0) != int(device):
            raise ValueError(f"catalog_device lives on {t.device}, the engine runs on cuda:{int(device)}")


def _topo_from_coord_mgr(coord_mgr, time_indices):
    """Topocentric unit vectors (len(time_indices), 3, nsrc) of every source at the given time indices
    from a matvis-style manager (``rotate(ti)``, attribute ``all_coords_topo``; ``setup()`` was called)."""
    out = []
    for ti in time_indices:
        coord_mgr.rotate(ti)
        out.append(np.array(coord_mgr.all_coords_topo, dtype=float))
    return np.stack(out)


def _time_block(device, nt, nf, nbls, polarized, precision, nsrc_topo=0):
    """Time steps per fv_sim_run: as many as keep the output block under 45 % of free device memory
    (and, with a coordinate manager, the staged vectors of a block under 2 GiB)."""
    from ..wrapper import device_memory_budget  # free + what this process's own cached handles hold

    per_time = max(1, nf * nbls * (4 if polarized else 1) * 8 * precision)
    n = max(1, int(0.45 * device_memory_budget(device) // per_time))
    if nsrc_topo:
        n = min(n, max(1, int(2**31 // (3 * nsrc_topo * 4 * precision))))
    return min(n, max(nt, 1))


if "fftvis.core.simulate" in __import__("sys").modules:
    register_with_reference()
