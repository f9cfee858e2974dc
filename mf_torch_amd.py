"""Import alias: ``import mf_torch_amd`` -> the package in ``matrix-factorization-torch_amd/``
(a hyphenated directory cannot be named in an ``import`` statement)."""
import importlib
import pathlib
import sys

_root = str(pathlib.Path(__file__).resolve().parent)
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("matrix-factorization-torch_amd")
