"""fftvis_amd -- MI355X-native GPU backend for the fftvis visibility simulator."""

__version__ = "0.1.0"
