"""Registers the import environment in which the reference's caller files can be executed in the build container (test
infrastructure; see README.md): the four operator shims of this directory, and EMPTY modules for third-party packages the
reference files import at module level but the executed code paths never call (plot / mesh / metric / GUI helpers).  An empty
stub that is ever touched raises AttributeError -- nothing is emulated."""
import importlib
import importlib.machinery
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

_EMPTY = ["trimesh", "trimesh.creation", "trimesh.primitives", "cv2", "imageio", "tensorboardX", "mcubes", "lpips", "torch_ema", "torchmetrics",
          "torchmetrics.functional", "json5", "pytorch3d", "pytorch3d.structures", "skspatial", "skspatial.objects", "open3d",
          "matplotlib", "matplotlib.pyplot"]
# names the reference binds with `from X import name`, or mentions in a signature annotation, at import time (placeholder classes,
# never used by the code the generator runs)
_NAMES = {"trimesh": ["Trimesh"], "trimesh.primitives": ["Box"], "trimesh.creation": ["uv_sphere"], "torch_ema": ["ExponentialMovingAverage"],
          "torchmetrics.functional": ["structural_similarity_index_measure"], "pytorch3d.structures": ["Meshes"], "pytorch3d": ["_C"],
          "skspatial.objects": ["Plane"]}


def install():
    sys.dont_write_bytecode = True           # never drop .pyc files into the read-only reference tree
    for name in _EMPTY:
        try:
            __import__(name)
            continue                          # the real package exists here: use it
        except Exception:
            pass
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        for attr in _NAMES.get(name, []):
            setattr(m, attr, type(attr, (), {}))
        sys.modules[name] = m
        if "." in name:
            setattr(sys.modules[name.rsplit(".", 1)[0]], name.rsplit(".", 1)[1], m)
    root = os.path.dirname(os.path.dirname(HERE))
    for p in (REF, root, HERE):               # final order: shims, repo root (`oracle`), reference
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    # bind the operator packages to the shims NOW: the reference's own `raymarching/` etc. JIT-build on import and must never load
    for name in ("raymarching", "gridencoder", "shencoder", "freqencoder"):
        mod = importlib.import_module(name)
        assert os.path.dirname(os.path.abspath(mod.__file__)).startswith(HERE), mod.__file__
