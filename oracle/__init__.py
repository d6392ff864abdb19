"""CPU oracle for the SealD-NeRF dynamic-NeRF rendering path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``seald-nerf_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / the timed baseline.
"""
from .oracle import *  # noqa: F401,F403
