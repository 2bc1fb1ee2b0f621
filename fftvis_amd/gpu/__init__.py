"""GPU-specific implementations (mirror of the reference's src/fftvis/gpu/__init__.py)."""

from .nufft import gpu_nufft2d, gpu_nufft3d  # noqa: F401
