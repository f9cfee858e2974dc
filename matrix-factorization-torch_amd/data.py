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


class DeviceInteractionSampler:
    """The training batches, produced on the GPU from HBM-resident interactions (``mf_sample_batch``):
    the counterpart of ``InteractionProcessor.get_batch_data`` (xfmr_rec/data/lightning.py:311-363) for
    id-only towers.  ``pair_user / pair_item / pair_target``: one entry per rating; the positive list
    of user u is ``pos_items[pos_off[u] : pos_off[u + 1]]`` (``UserProcessor.process``, :274-280).
    ``batch(step)`` is example positions ``step * batch_size ...`` of a stream that is reshuffled
    every epoch; the same (seed, step) always gives the same batch."""

    def __init__(self, pair_user, pair_item, pair_target, pos_off, pos_items, *, num_items: int,
                 batch_size: int = BATCH_SIZE, pos_pad: int = 64, seed: int = 0, device="cuda") -> None:
        from . import _lib

        self._lib = _lib
        i64 = lambda t: torch.as_tensor(t, dtype=torch.int64).to(device).contiguous()  # noqa: E731
        self.pair_user, self.pair_item = i64(pair_user), i64(pair_item)
        self.pair_target = torch.as_tensor(pair_target, dtype=torch.float32).to(device).contiguous()
        self.pos_off, self.pos_items = i64(pos_off), i64(pos_items)
        if self.pos_items.numel() == 0:
            self.pos_items = torch.zeros(1, dtype=torch.int64, device=device)
        self.num_items, self.batch_size, self.pos_pad, self.seed = int(num_items), int(batch_size), int(pos_pad), int(seed)
        n = self.pair_user.numel()
        if not (self.pair_item.numel() == n == self.pair_target.numel()) or n == 0:
            msg = "pair_user, pair_item and pair_target must have the same, non-zero length"
            raise ValueError(msg)

    @property
    def steps_per_epoch(self) -> int:
        return -(-self.pair_user.numel() // self.batch_size)

    def batch(self, step: int) -> InteractionBatchType:
        b, p, dev = self.batch_size, self.pos_pad, self.pair_user.device
        user = torch.empty(b, dtype=torch.int64, device=dev)
        item = torch.empty(2 * b, dtype=torch.int64, device=dev)
        target = torch.empty(b, dtype=torch.float32, device=dev)
        pos = torch.empty(b, p, dtype=torch.int64, device=dev)
        lib = self._lib
        lib.check(lib.lib().mf_sample_batch(self.pair_user.data_ptr(), self.pair_item.data_ptr(), self.pair_target.data_ptr(),
                                            self.pair_user.numel(), self.pos_off.data_ptr(), self.pos_items.data_ptr(),
                                            self.num_items, self.seed, int(step) * b, b, p, user.data_ptr(), item.data_ptr(),
                                            target.data_ptr(), pos.data_ptr(), lib.stream_ptr()))
        return {"target": target, "user": {"idx": user, "pos_idx": pos}, "item": {"idx": item[:b]},
                "neg_item": {"idx": item[b:]}}

    def __iter__(self) -> Iterator[InteractionBatchType]:
        step = 0
        while True:
            yield self.batch(step)
            step += 1


def to_device(batch, device):
    if isinstance(batch, dict):
        return {k: to_device(v, device) for k, v in batch.items()}
    return batch.to(device) if isinstance(batch, torch.Tensor) else batch
