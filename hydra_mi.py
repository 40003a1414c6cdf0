"""Importable alias of the package directory ``kalman-hydra_amd/``.

The build contract fixes the directory name, which is not a valid Python
identifier; ``import hydra_mi`` (or ``importlib.import_module("kalman-hydra_amd")``)
gives the same module objects, including for ``from hydra_mi.<sub> import ...``.
"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("kalman-hydra_amd")
for _sub in ("_lib", "matio", "synth", "mesh", "brox", "renderer", "kalman", "pipeline", "batch", "imgproc", "distmesh_dyn"):
    sys.modules[__name__ + "." + _sub] = importlib.import_module("kalman-hydra_amd." + _sub)
sys.modules[__name__] = _pkg
