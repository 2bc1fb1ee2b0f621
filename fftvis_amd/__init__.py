"""fftvis_amd -- MI355X-native GPU backend for the fftvis visibility simulator.

Mirrors the reference package's public surface for the gpu backend
(src/fftvis/__init__.py:1-31): ``simulate_vis``, the engine / evaluator factories and the
``gpu`` sub-package.
"""

__version__ = "0.1.0"

from .core.beams import AiryBeam, TabulatedBeam  # noqa: F401
from .core.beam_basis import compute_beam_basis, compute_beam_basis_per_freq  # noqa: F401
from .core.simulate import SimulationEngine, default_accuracy_dict  # noqa: F401
from .wrapper import create_beam_evaluator, create_simulation_engine, simulate_vis  # noqa: F401
