"""Worker of tests/test_gpu_scale.py::test_fft_layout_switches_leave_the_result_bit_identical: one seeded 2-D
type-3 transform through the GPU library under whatever FFTVIS_HIP_* switches the parent set; saves the result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fftvis_amd.gpu import gpu_nufft2d  # noqa: E402


def problem():
    rng = np.random.default_rng(17)
    M, N = 500, 300
    x, y = rng.uniform(-3, 3, (2, M))
    c = rng.normal(size=(3, M)) + 1j * rng.normal(size=(3, M))
    s = rng.uniform(-1040, 1040, N)   # 8192 x 6144 fine grid: both passes on the register-resident kernels,
    t = rng.uniform(-700, 700, N)     # the y-pass folded (3 x 2048)
    return x, y, c, s, t


def problem_folded():
    """C3's widest grid, 10240 x 8192 = (5 x 2048) x (4 x 2048): both passes folded, odd and even residue counts."""
    rng = np.random.default_rng(18)
    M, N = 500, 300
    x, y = rng.uniform(-3, 3, (2, M))
    c = rng.normal(size=(3, M)) + 1j * rng.normal(size=(3, M))
    return x, y, c, rng.uniform(-1300, 1300, N), rng.uniform(-1040, 1040, N)


if __name__ == "__main__":
    which = problem_folded if len(sys.argv) > 2 and sys.argv[2] == "folded" else problem
    np.save(sys.argv[1], gpu_nufft2d(*which(), 1e-9))
