"""CPU oracle for the fftvis hot path: test infrastructure only (see fftvis_oracle.py)."""
