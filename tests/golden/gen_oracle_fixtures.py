#!/usr/bin/env python3
"""Writes tests/golden/oracle_*.npz: bit-level regression pins of the CPU oracle on small seeded cases
(the cases live in tests/tests_support.py).  These pin the ORACLE against accidental change; they are
not reference outputs (the reference's CUDA kernels cannot run here -- see README.md)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "seald-nerf_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from tests_support import oracle_fixture_cases  # noqa: E402

if __name__ == "__main__":
    for name, fn in oracle_fixture_cases().items():
        np.savez_compressed(os.path.join(HERE, f"oracle_{name}.npz"), **fn())
        print("wrote", name)
