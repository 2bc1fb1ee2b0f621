"""GPU-specific implementations (mirror of the reference's src/fftvis/gpu/__init__.py)."""

from .beams import GPUBeamEvaluator  # noqa: F401
from .gpu_simulate import GPUSimulationEngine  # noqa: F401
from .nufft import gpu_nufft2d, gpu_nufft3d  # noqa: F401
