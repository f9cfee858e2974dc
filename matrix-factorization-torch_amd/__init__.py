"""matrix-factorization-torch_amd: the MI355X (gfx950) hot path of ``xfmr_rec``.

Embedding-table towers, the seven in-batch losses of ``xfmr_rec/losses.py`` (forward
and backward), sparse SGD / row-wise Adam updates and exact brute-force top-k
retrieval, as hand-written HIP kernels behind the reference's Python call surface.
See DESIGN.md; C ABI in include/mf_hip.h.

The directory name has a hyphen (fixed by the project layout), so import it with
``importlib.import_module("matrix-factorization-torch_amd")`` or through the
``mf_torch_amd`` alias module at the repository root.
"""
from . import _lib, data, distributed, fused, graph, lightning, losses, models, optim, params, retrieval  # noqa: F401
from ._lib import MfHipError, build  # noqa: F401

__all__ = ["data", "distributed", "fused", "graph", "lightning", "losses", "models", "optim", "params", "retrieval", "MfHipError", "build"]
