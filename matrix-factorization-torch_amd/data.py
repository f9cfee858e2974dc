"""Batch layout of the hot path (the ids-only part of ``xfmr_rec/data``).

What the loss path consumes is fixed by the reference's collate
(``InteractionProcessor.collate``, xfmr_rec/data/lightning.py:335-342, batch type
``InteractionBatchType`` :72-76): ``target[B]`` int64 ratings, ``user.pos_idx[B, P]``
int64 right-padded with 0 (``pad_tensors`` / ``collate_tensor_fn``,
xfmr_rec/data/load.py:38-75), ``item.idx[B]`` 1-based item rows (``movie_rn``,
prepare.py:85) and ``neg_item.idx[B]`` uniform negatives (:344-354).  The towers here
are embedding tables, so ``user`` carries ``idx`` (``user_rn``) where the reference
carries ``text``.

The reference's ETL (polars / parquet / LanceDB, prepare.py, load.py:78-141) is out of
scope; :class:`SyntheticInteractions` produces MovieLens-*shaped*
batches (no MovieLens files are available offline) for tests and benchmarks.
"""
from __future__ import annotations

from typing import Iterable, Iterator, TypedDict

import torch
import torch.nn.functional as F  # noqa: N812

from .params import BATCH_SIZE, PADDING_IDX


class ItemBatchType(TypedDict):
    idx: torch.Tensor


class UserBatchType(TypedDict):
    idx: torch.Tensor
    pos_idx: torch.Tensor


class InteractionBatchType(TypedDict):
    target: torch.Tensor
    user: UserBatchType
    item: ItemBatchType
    neg_item: ItemBatchType


def pad_tensors(batch: Iterable[torch.Tensor], dim: int = -1, *, pad_start: bool = False,
                pad_value: int = PADDING_IDX) -> torch.Tensor:
    """Stack ragged tensors, padding ``dim`` to the longest (same contract as load.py:38-55,
    whose 12 shape cases are replayed in tests/test_data.py)."""
    tensors = list(batch)
    size = max(t.size(dim) for t in tensors)
    nd = tensors[0].dim()
    axis = dim % nd
    out = []
    for t in tensors:
        pad = [0, 0] * nd
        # F.pad lists dimensions last-to-first; [2j] pads before, [2j + 1] after
        pad[2 * (nd - 1 - axis) + (0 if pad_start else 1)] = size - t.size(dim)
        out.append(F.pad(t, pad, value=pad_value))
    return torch.stack(out)


def collate_interactions(examples: list[dict]) -> InteractionBatchType:
    """List of ``{"target", "user": {"idx", "pos_idx"}, "item": {"idx"}, "neg_item": {"idx"}}``
    -> batch (ragged ``pos_idx`` rows are 0-padded on the right)."""
    as_t = lambda xs: torch.as_tensor(xs, dtype=torch.int64)  # noqa: E731
    return {
        "target": as_t([e["target"] for e in examples]),
        "user": {
            "idx": as_t([e["user"]["idx"] for e in examples]),
            "pos_idx": pad_tensors([as_t(e["user"]["pos_idx"]) for e in examples], dim=-1),
        },
        "item": {"idx": as_t([e["item"]["idx"] for e in examples])},
        "neg_item": {"idx": as_t([e["neg_item"]["idx"] for e in examples])},
    }


class SyntheticInteractions:
    """MovieLens-shaped synthetic interactions: Zipf(s) item popularity, log-normal user
    activity, ratings uniform in 1..5, each user's positives a fixed random item set
    (SURVEY.md 8d).  Rows are 1-based; row 0 is the padding row of both tables."""

    def __init__(self, num_users: int, num_items: int, *, max_positives: int = 64, zipf_s: float = 1.0,
                 seed: int = 0, user_range: tuple[int, int] | None = None) -> None:
        self.num_users, self.num_items, self.max_positives = num_users, num_items, max_positives
        self.gen = torch.Generator().manual_seed(seed)
        self.item_w = 1.0 / torch.arange(1, num_items, dtype=torch.float64) ** zipf_s
        lo, hi = user_range if user_range is not None else (1, num_users)
        self.user_lo = lo
        self.user_w = torch.exp(torch.randn(hi - lo, generator=self.gen, dtype=torch.float64))

    def item_probability(self) -> torch.Tensor:
        """Sampling probability of each item row as a batch column (positives ~ Zipf, negatives ~ uniform)."""
        p = 0.5 * self.item_w / self.item_w.sum() + 0.5 / (self.num_items - 1)
        return torch.cat([torch.zeros(1, dtype=torch.float64), p]).to(torch.float32)

    def batch(self, batch_size: int = BATCH_SIZE) -> InteractionBatchType:
        g, p = self.gen, self.max_positives
        user = torch.multinomial(self.user_w, batch_size, replacement=True, generator=g) + self.user_lo
        item = torch.multinomial(self.item_w, batch_size, replacement=True, generator=g) + 1
        neg = torch.randint(1, self.num_items, (batch_size,), generator=g)       # uniform negatives (:344-354)
        target = torch.randint(1, 6, (batch_size,), generator=g)
        n_pos = torch.randint(min(8, p), p + 1, (batch_size,), generator=g)
        pos = torch.multinomial(self.item_w, batch_size * p, replacement=True, generator=g).reshape(batch_size, p) + 1
        pos[:, 0] = item
        pos[torch.arange(p)[None, :] >= n_pos[:, None]] = PADDING_IDX
        return {"target": target, "user": {"idx": user, "pos_idx": pos}, "item": {"idx": item},
                "neg_item": {"idx": neg}}

    def __iter__(self) -> Iterator[InteractionBatchType]:
        while True:
            yield self.batch()


def to_device(batch, device):
    if isinstance(batch, dict):
        return {k: to_device(v, device) for k, v in batch.items()}
    return batch.to(device) if isinstance(batch, torch.Tensor) else batch
