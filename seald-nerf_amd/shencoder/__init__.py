"""Drop-in for the reference's `shencoder` package (shencoder/sphere_harmonics.py), backed by libsdn_hip."""
from .sphere_harmonics import SHEncoder, sh_encode  # noqa: F401
