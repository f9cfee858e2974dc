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
    num_hashes: int = 0            # > 0: hash / bloom towers (config 5): num_users / num_items are BUCKET counts
    hash_seed: int = 0


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


class _GatherHashed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table: torch.Tensor, idx: torch.Tensor, num_hashes: int, seed: int, normalize: bool):
        if not table.is_cuda or table.dtype != torch.float32 or not table.is_contiguous():
            raise _lib.MfHipError("embedding table must be a contiguous fp32 tensor on the GPU")
        ids = _lib.dev_i64(idx, "idx").reshape(-1)
        n, d = ids.numel(), table.shape[1]
        out = torch.empty(n, d, dtype=torch.float32, device=table.device)
        inv = torch.empty(n, dtype=torch.float32, device=table.device)
        _lib.check(_lib.lib().mf_gather_hashed(table.data_ptr(), table.shape[0], d, ids.data_ptr(), n, num_hashes, seed,
                                               int(normalize), out.data_ptr(), inv.data_ptr(), _lib.stream_ptr()))
        ctx.table, ctx.ids, ctx.cfg = table, ids, (num_hashes, seed, bool(normalize))
        ctx.save_for_backward(out, inv)
        return out.reshape(*idx.shape, d)

    @staticmethod
    def backward(ctx, grad_out):
        table, ids = ctx.table, ctx.ids
        num_hashes, seed, normalize = ctx.cfg
        out, inv = ctx.saved_tensors
        n, d = ids.numel(), table.shape[1]
        lib = _lib.lib()
        g = grad_out.reshape(-1, d).to(torch.float32).contiguous()
        if normalize:      # gradient w.r.t. the summed rows: the same for each of the id's bucket rows
            graw = torch.empty_like(g)
            _lib.check(lib.mf_normalize_backward(out.data_ptr(), inv.data_ptr(), g.data_ptr(), n, d, graw.data_ptr(),
                                                 _lib.stream_ptr()))
            g = graw
        buckets = torch.empty(n * num_hashes, dtype=torch.int64, device=table.device)
        _lib.check(lib.mf_hash_buckets(ids.data_ptr(), n, num_hashes, seed, table.shape[0], buckets.data_ptr(),
                                       _lib.stream_ptr()))
        pending = getattr(table, "_mf_pending", None)
        if pending is None:
            pending = []
            table._mf_pending = pending
        pending.append((buckets, g.repeat_interleave(num_hashes, dim=0), False))
        return None, None, None, None, None


class HashEmbeddingTower(torch.nn.Module):
    """Hash / bloom embedding tower (BASELINE config 5: 100 M items x 10 M users do not get a row
    each): ``tower(idx)`` = L2-normalised sum of the ``num_hashes`` bucket rows of every id
    (``include/mf_numerics.h`` ``mf_hash_bucket``; spec ``oracle/embed.py``, no reference
    counterpart).  Works with the same sparse optimisers: the backward parks the bucket rows."""

    def __init__(self, num_buckets: int, embedding_dim: int, *, num_hashes: int = 2, seed: int = 0,
                 normalize: bool = True, init_std: float | None = None, device=None) -> None:
        super().__init__()
        if embedding_dim not in _lib.SUPPORTED_WIDTHS:
            msg = f"embedding_dim must be one of {_lib.SUPPORTED_WIDTHS}: {embedding_dim = }"
            raise ValueError(msg)
        if not 1 <= num_hashes <= 4:  # noqa: PLR2004
            msg = f"num_hashes must be in 1..4: {num_hashes = }"
            raise ValueError(msg)
        std = init_std if init_std is not None else 1.0 / math.sqrt(embedding_dim * num_hashes)
        self.weight = torch.nn.Parameter(torch.randn(num_buckets, embedding_dim, device=device) * std)
        self.num_hashes, self.seed, self.normalize = num_hashes, int(seed), normalize

    def forward(self, idx: torch.Tensor) -> torch.Tensor:
        return _GatherHashed.apply(self.weight, idx, self.num_hashes, self.seed, self.normalize)

    def buckets(self, idx: torch.Tensor) -> torch.Tensor:
        """``[*, num_hashes]`` table rows of every id."""
        ids = _lib.dev_i64(idx, "idx").reshape(-1)
        out = torch.empty(ids.numel() * self.num_hashes, dtype=torch.int64, device=ids.device)
        _lib.check(_lib.lib().mf_hash_buckets(ids.data_ptr(), ids.numel(), self.num_hashes, self.seed, self.weight.shape[0],
                                              out.data_ptr(), _lib.stream_ptr()))
        return out.reshape(*idx.shape, self.num_hashes)

    def extra_repr(self) -> str:
        return f"{self.weight.shape[0]} buckets, {self.weight.shape[1]}, num_hashes={self.num_hashes}, normalize={self.normalize}"


def init_towers(config: ModelConfig, device=None) -> torch.nn.ModuleDict:
    """Counterpart of ``init_bert`` + ``to_sentence_transformer`` (models.py:27-63)."""
    if config.num_hashes > 0:
        return torch.nn.ModuleDict(
            {
                name: HashEmbeddingTower(rows, config.hidden_size, num_hashes=config.num_hashes,
                                         seed=config.hash_seed + salt, normalize=config.normalize,
                                         init_std=config.init_std, device=device)
                for salt, (name, rows) in enumerate((("user", config.num_users), ("item", config.num_items)))
            }
        )
    return torch.nn.ModuleDict(
        {
            "user": EmbeddingTower(config.num_users, config.hidden_size, normalize=config.normalize,
                                   init_std=config.init_std, device=device),
            "item": EmbeddingTower(config.num_items, config.hidden_size, normalize=config.normalize,
                                   init_std=config.init_std, device=device),
        }
    )
