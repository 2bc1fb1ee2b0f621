"""ctypes binding of libfftvis_hip.so (C ABI declared in include/fftvis_hip.h).

The HIP library is the only compute path of this package: if it cannot be loaded the
import of any compute entry point raises -- there is no CPU fallback.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FFTVIS_HIP_LIB: another build of the same library (A/B measurements of kernel variants), else the in-tree one
LIB_PATH = os.environ.get("FFTVIS_HIP_LIB") or os.path.join(_HERE, "libfftvis_hip.so")
CSRC = os.path.join(_HERE, "csrc")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
    "-Wall", "-Wno-unused-function",
]

# name -> (restype, argtypes): every symbol include/fftvis_hip.h declares.
SYMBOLS = {
    "fv_version": (c_int, []),
    "fv_device_count": (c_int, [POINTER(c_int)]),
    "fv_last_error": (c_char_p, []),
    "fv_device_bytes": (c_int, [POINTER(c_int64)]),
    "fv_device_bytes_on": (c_int, [c_int, POINTER(c_int64)]),
    "fv_device_mem_info": (c_int, [c_int, POINTER(c_int64), POINTER(c_int64)]),
    "fv_release_workspaces": (c_int, []),
    "fv_nufft3": (c_int, [c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_int, c_int64, c_void_p, c_void_p, c_void_p, c_double, c_double,
                          c_void_p]),
    "fv_nudft3_direct": (c_int, [c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fv_beam_eval": (c_int, [c_int, c_int, c_int, c_int, c_double, c_int, c_int, c_int, c_double,
                             c_void_p, c_int, c_int, c_double, c_int64, c_void_p, c_void_p, c_void_p]),
    "fv_apparent_coherency": (c_int, [c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "fv_inplace_rot": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int64]),
    "fv_astrom_topo": (c_int, [c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "fv_sim_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_double, c_double, c_int]),
    "fv_sim_destroy": (c_int, [c_void_p]),
    "fv_sim_set_sources": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int]),
    "fv_sim_set_times": (c_int, [c_void_p, c_int, c_void_p]),
    "fv_sim_set_topo": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int]),
    "fv_sim_set_astrom": (c_int, [c_void_p, c_int, c_void_p]),
    "fv_sim_set_freqs": (c_int, [c_void_p, c_int, c_void_p]),
    "fv_sim_set_array": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int]),
    "fv_sim_set_array_type1": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int]),
    "fv_sim_set_nbeams": (c_int, [c_void_p, c_int]),
    "fv_sim_set_beam_airy": (c_int, [c_void_p, c_int, c_double]),
    "fv_sim_set_beam_airy_scaled": (c_int, [c_void_p, c_int, c_double, c_void_p, c_double]),
    "fv_sim_set_reference_compat": (c_int, [c_void_p, c_int]),
    "fv_sim_set_beam_table": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p, c_int]),
    "fv_sim_set_beam_pairs": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "fv_sim_set_basis": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fv_sim_set_chunking": (c_int, [c_void_p, c_int, c_double]),
    "fv_sim_run": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int]),
    "fv_sim_run_into": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int]),
    "fv_comm_unique_id": (c_int, [c_void_p]),
    "fv_comm_init": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_void_p]),
    "fv_comm_destroy": (c_int, [c_void_p]),
    "fv_bcast_catalog": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64]),
    "fv_scatter_flux_columns": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_void_p, POINTER(c_int), c_void_p]),
    "fv_sim_sync": (c_int, [c_void_p]),
    "fv_sim_stats": (c_int, [c_void_p, c_void_p, c_int]),
    "fv_sim_reset_stats": (c_int, [c_void_p]),
    "fv_sim_enable_timing": (c_int, [c_void_p, c_int]),
    "fv_sim_timing": (c_int, [c_void_p, c_void_p, c_int]),
}


class FftvisHipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile libfftvis_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    hdr = os.path.join(os.path.dirname(_HERE), "include", "fftvis_hip.h")
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(p) for p in srcs + [hdr])
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("FFTVIS_HIP_EXTRA_FLAGS", "").split()
    cmd = [hipcc, *HIPCC_FLAGS, *extra, os.path.join(CSRC, "fv_capi.hip"), "-o", LIB_PATH]
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def _preload_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own ROCm runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7, and
    libhsa-runtime64.so) and load it by path; libfftvis_hip.so asks for the soname and, loaded FIRST, gets the system's
    /opt/rocm copy.  A later ``import torch`` then brings a second HIP runtime into the process, and the second one
    sees no GPU ("No HIP GPUs are available"; RCCL: "unhandled cuda error").  Loading torch's copy first -- the file,
    without importing torch -- makes either import order end with ONE runtime: ours binds to the loaded soname, torch's
    later load of the same file is a no-op.  FFTVIS_HIP_NO_TORCH_RUNTIME=1 leaves it out."""
    if os.environ.get("FFTVIS_HIP_NO_TORCH_RUNTIME"):
        return
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # already in: its runtime is the loaded one
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return  # (a wheel whose runtime does not load by itself: leave the system's in charge)


def lib() -> ctypes.CDLL:
    """Load the library (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FftvisHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  fftvis_amd has no CPU fallback."
            )
        _preload_torch_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != 0:
        msg = lib().fv_last_error()
        raise FftvisHipError(f"libfftvis_hip status {status}: {msg.decode() if msg else ''}")


def device_count() -> int:
    n = c_int(0)
    check(lib().fv_device_count(ctypes.byref(n)))
    return n.value


def require_gpu() -> None:
    if device_count() < 1:
        raise FftvisHipError("no HIP device visible: the fftvis_amd gpu backend needs an MI355X")


def ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(c_void_p)
    return c_void_p(int(a))  # raw device pointer
