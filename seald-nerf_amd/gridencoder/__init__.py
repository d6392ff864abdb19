"""Drop-in for the reference's `gridencoder` package (gridencoder/grid.py), backed by libsdn_hip."""
from .grid import GridEncoder, grid_encode  # noqa: F401
