"""Drop-in for the reference's `ffmlp` package (ffmlp/ffmlp.py), backed by libsdn_hip."""
from .ffmlp import FFMLP, ffmlp_forward, convert_activation  # noqa: F401
