"""Towers: embedding tables instead of the reference's text encoder.

``xfmr_rec/models.py`` builds ONE randomly initialised BERT wrapped as
Transformer -> mean-Pooling -> Normalize (models.py:27-63) and the Lightning module
encodes user / item JSON text with it (xfmr_rec/lightning.py:60-74).  The north-star
of this repository replaces that tower by user / item **embedding tables**
(``ModelConfig`` keeps the reference's field names where they still mean something):
the forward is a coalesced row gather on the GPU (``mf_gather_rows``), ending in the
same L2-normalisation, and the backward does not build a dense table gradient: it
parks ``(row ids, row gradients)`` on the parameter for the sparse optimisers of
``optim.py``.  No reference implementation exists for this part; the
spec is ``oracle/embed.py``.
"""
from __future__ import annotations

import math

import pydantic
import torch

from . import _lib


class ModelConfig(pydantic.BaseModel):
    """Fields of ``xfmr_rec.models.ModelConfig`` (models.py:14-24) that survive the
    tower swap, plus the table sizes.  ``hidden_size`` is the embedding width d."""

    num_users: int = 6041          # ML-1M: 6,040 users + padding row 0
    num_items: int = 3884          # ML-1M: 3,883 items + padding row 0 (idx are 1-based, prepare.py:85)
    hidden_size: int = 64
    normalize: bool = True         # models.Normalize() at the end of the tower (models.py:59)
    init_std: float | None = None  # default 1/sqrt(hidden_size)


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table: torch.Tensor, idx: torch.Tensor, normalize: bool):
        if not table.is_cuda or table.dtype != torch.float32 or not table.is_contiguous():
            raise _lib.MfHipError("embedding table must be a contiguous fp32 tensor on the GPU")
        ids = _lib.dev_i64(idx, "idx").reshape(-1)
        n, d = ids.numel(), table.shape[1]
        out = torch.empty(n, d, dtype=torch.float32, device=table.device)
        _lib.check(_lib.lib().mf_gather_rows(table.data_ptr(), table.shape[0], d, ids.data_ptr(), n, int(normalize),
                                             out.data_ptr(), None, _lib.stream_ptr()))
        ctx.table = table
        ctx.ids = ids
        ctx.normalize = bool(normalize)
        return out.reshape(*idx.shape, d)

    @staticmethod
    def backward(ctx, grad_out):
        table = ctx.table
        g = grad_out.reshape(-1, table.shape[1]).to(torch.float32).contiguous()
        pending = getattr(table, "_mf_pending", None)
        if pending is None:
            pending = []
            table._mf_pending = pending
        pending.append((ctx.ids, g, ctx.normalize))
        return None, None, None


class EmbeddingTower(torch.nn.Module):
    """``tower(idx) -> [*, d]`` unit-norm rows; drop-in for ``module(text)``."""

    def __init__(self, num_embeddings: int, embedding_dim: int, *, normalize: bool = True,
                 init_std: float | None = None, device=None) -> None:
        super().__init__()
        if embedding_dim not in _lib.SUPPORTED_WIDTHS:
            msg = f"embedding_dim must be one of {_lib.SUPPORTED_WIDTHS}: {embedding_dim = }"
            raise ValueError(msg)
        std = init_std if init_std is not None else 1.0 / math.sqrt(embedding_dim)
        w = torch.randn(num_embeddings, embedding_dim, device=device) * std
        self.weight = torch.nn.Parameter(w)
        self.normalize = normalize

    @property
    def num_embeddings(self) -> int:
        return self.weight.shape[0]

    @property
    def embedding_dim(self) -> int:
        return self.weight.shape[1]

    def forward(self, idx: torch.Tensor) -> torch.Tensor:
        return _GatherRows.apply(self.weight, idx, self.normalize)

    def extra_repr(self) -> str:
        return f"{self.num_embeddings}, {self.embedding_dim}, normalize={self.normalize}"


def init_towers(config: ModelConfig, device=None) -> torch.nn.ModuleDict:
    """Counterpart of ``init_bert`` + ``to_sentence_transformer`` (models.py:27-63)."""
    return torch.nn.ModuleDict(
        {
            "user": EmbeddingTower(config.num_users, config.hidden_size, normalize=config.normalize,
                                   init_std=config.init_std, device=device),
            "item": EmbeddingTower(config.num_items, config.hidden_size, normalize=config.normalize,
                                   init_std=config.init_std, device=device),
        }
    )
