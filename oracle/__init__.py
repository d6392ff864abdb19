"""CPU oracle for the SealD-NeRF dynamic-NeRF rendering path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``seald-nerf_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / the timed baseline.

Pinned by reference-generated fixtures (tests/golden/README.md): per-operator vectors and, since round 2, the outputs of the
reference's own caller code executed over these operators (caller_*.npz).  ``ffmlp`` alone is "parity unpinned".
"""
from .oracle import *  # noqa: F401,F403
