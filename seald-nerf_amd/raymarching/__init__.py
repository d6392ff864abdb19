"""Drop-in for the reference's `raymarching` package (raymarching/raymarching.py), backed by libsdn_hip."""
from .raymarching import *  # noqa: F401,F403
