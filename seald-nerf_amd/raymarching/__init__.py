"""Drop-in for the reference's `raymarching` package (raymarching/raymarching.py), backed by libsdn_hip."""
from .raymarching import *  # noqa: F401,F403
from .raymarching import _cull_grid_of  # noqa: F401  (tests look at which marcher a slice takes)
from .raymarching import live_lists  # noqa: F401  (the switch of the marcher's live-sample lists: one dict, shared)
