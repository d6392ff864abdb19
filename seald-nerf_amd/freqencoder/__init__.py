"""Drop-in for the reference's `freqencoder` package (freqencoder/freq.py), backed by libsdn_hip."""
from .freq import FreqEncoder, freq_encode  # noqa: F401
