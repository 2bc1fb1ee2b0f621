"""Backend-independent host helpers (mirror of the reference's src/fftvis/core/)."""
