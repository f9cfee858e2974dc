"""Batch layout of the hot path (the ids-only part of ``xfmr_rec/data``).

What the loss path consumes is fixed by the reference's collate
(``InteractionProcessor.collate``, xfmr_rec/data/lightning.py:335-342, batch type
``InteractionBatchType`` :72-76): ``target[B]`` int64 ratings, ``user.pos_idx[B, P]``
int64 right-padded with 0 (``pad_tensors`` / ``collate_tensor_fn``,
xfmr_rec/data/load.py:38-75), ``item.idx[B]`` 1-based item rows (``movie_rn``,
prepare.py:85) and ``neg_item.idx[B]`` uniform negatives (:344-354).  The towers here
are embedding tables, so ``user`` carries ``idx`` (``user_rn``) where the reference
carries ``text``.

The reference's text ETL (polars / parquet / LanceDB, prepare.py, load.py:78-141) is out of
scope; its ids-only core is here: :func:`split_ratings` (the per-user temporal
80/20 split and the val / test user split, prepare.py:160-194), :class:`InteractionTable` (train
pairs, per-user train positives, the 4-week rolling history of every rating, prepare.py:229-243, and
the evaluation sets of ``process_users``, :272-310) and :class:`DeviceInteractionSampler` (the batch
producer on the GPU).  :class:`SyntheticInteractions` produces MovieLens-*shaped* batches (no MovieLens
files are available offline) for tests and benchmarks.
"""
from __future__ import annotations

from typing import Iterable, Iterator, TypedDict

import torch
import torch.nn.functional as F  # noqa: N812

from .params import BATCH_SIZE, PADDING_IDX


class ItemBatchType(TypedDict):
    idx: torch.Tensor


class UserBatchType(TypedDict, total=False):
    idx: torch.Tensor
    pos_idx: torch.Tensor                 # the reference's padded positives [B, P] ...
    pos_csr: tuple                        # ... or (user idx [B], pos_off [U + 1], pos_items): the lists in place (no padding)


class InteractionBatchType(TypedDict):
    target: torch.Tensor
    user: UserBatchType
    item: ItemBatchType
    neg_item: ItemBatchType


def pad_tensors(batch: Iterable[torch.Tensor], dim: int = -1, *, pad_start: bool = False,
                pad_value: int = PADDING_IDX) -> torch.Tensor:
    """Stack ragged tensors, padding ``dim`` to the longest (same contract as load.py:38-55,
    whose 12 shape cases are replayed in tests/test_host_cpu.py)."""
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


FOUR_WEEKS = 4 * 7 * 24 * 3600       # polars "4w", in the seconds of a MovieLens timestamp


def _segment_layout(keys_sorted: torch.Tensor):
    """(segment id of every position, first position of every segment, length of every segment) of a sorted key vector."""
    n = keys_sorted.numel()
    new = torch.ones(n, dtype=torch.bool, device=keys_sorted.device)
    new[1:] = keys_sorted[1:] != keys_sorted[:-1]
    seg = torch.cumsum(new.to(torch.int64), 0) - 1
    start = torch.nonzero(new).flatten()
    length = torch.diff(torch.cat([start, torch.tensor([n], device=start.device)]))
    return seg, start, length


def _rank_min(values_sorted: torch.Tensor, group_start_of_pos: torch.Tensor) -> torch.Tensor:
    """polars ``rank("min")`` (1-based; ties share the lowest rank) of values that are sorted inside their groups:
    position of the first element of the tie run, relative to its group's first position."""
    n = values_sorted.numel()
    pos = torch.arange(n, device=values_sorted.device)
    run_new = torch.ones(n, dtype=torch.bool, device=values_sorted.device)
    run_new[1:] = (values_sorted[1:] != values_sorted[:-1]) | (group_start_of_pos[1:] != group_start_of_pos[:-1])
    run_start = torch.cummax(torch.where(run_new, pos, torch.zeros_like(pos)), 0).values
    return run_start - group_start_of_pos + 1


def split_ratings(user: torch.Tensor, timestamp: torch.Tensor, *, train_prop: float = 0.8, val_prop: float = 0.2):
    """``train_test_split`` of the reference (xfmr_rec/data/prepare.py:160-194) on id / time vectors: per user the first
    ``train_prop`` of its ratings IN TIME are train (``p = (rank_min(datetime) - 1) / count < train_prop``); of the users
    that own non-train ratings, those with the largest ``val_prop`` share of such ratings (``(rank_min(len) - 1) /
    n_users >= 1 - val_prop``) are validation users, the others test users.  Returns bool vectors
    ``is_train, is_val, is_test`` aligned with the input."""
    n = user.numel()
    o1 = torch.argsort(timestamp, stable=True)
    order = o1[torch.argsort(user[o1], stable=True)]                 # by (user, time), ties in input order
    su, st = user[order], timestamp[order]
    seg, start, length = _segment_layout(su)
    rank = _rank_min(st, start[seg])
    p = (rank - 1).to(torch.float64) / length[seg].to(torch.float64)
    train_sorted = p < train_prop
    is_train = torch.empty(n, dtype=torch.bool, device=user.device)
    is_train[order] = train_sorted
    # users_split: over the users with at least one non-train rating
    nontrain = torch.zeros(start.numel(), dtype=torch.int64, device=user.device)
    nontrain.index_add_(0, seg, (~train_sorted).to(torch.int64))
    has = nontrain > 0
    lens = nontrain[has]
    lo = torch.argsort(lens, stable=True)
    r = _rank_min(lens[lo], torch.zeros_like(lens))
    pv = torch.empty(lens.numel(), dtype=torch.float64, device=user.device)
    pv[lo] = (r - 1).to(torch.float64) / max(lens.numel(), 1)
    val_user = torch.zeros(start.numel(), dtype=torch.bool, device=user.device)
    val_user[has] = pv >= 1.0 - val_prop
    val_sorted = (~train_sorted) & val_user[seg]
    is_val = torch.empty_like(is_train)
    is_val[order] = val_sorted
    return is_train, is_val, (~is_train) & (~is_val)


class InteractionTable:
    """The ids-only interaction tables a training / evaluation run needs, built once from the ratings
    (``user``, ``item`` -- the 1-based ``movie_rn`` -- , ``rating``, ``timestamp`` in seconds), all torch ops (the tensors
    may live on the GPU).  Follows ``prepare_movielens`` (prepare.py:313-325) for everything that is not text:

    * ``is_train / is_val / is_test`` -- :func:`split_ratings`;
    * ``pair_user / pair_item / pair_target`` -- the train ratings (``ratings.parquet`` filtered by ``is_train``,
      data/lightning.py:344-354), sorted by (user, time);
    * ``pos_off / pos_items`` -- every user's train items in time order: the ``target`` of a train row
      (``gather_history``, prepare.py:236-241) whose item rows become ``pos_idx`` (``UserProcessor.process``,
      data/lightning.py:274-280);
    * ``history_lo / history_hi`` -- per rating (in (user, time) order, ``order`` maps back to the input): its rolling
      4-week history is ``sorted_item[history_lo : history_hi]`` -- the user's ratings with ``t - 4w < t' < t``
      (``rolling("datetime", period="4w", closed="none")``, prepare.py:230-234);
    * ``eval_sets(split)`` -- per user of the split (``process_users``, :272-310): ``history`` = its train items
      (excluded from retrieval), ``target`` = its non-train items with their ratings."""

    def __init__(self, user, item, rating, timestamp, *, train_prop: float = 0.8, val_prop: float = 0.2) -> None:
        user, item = torch.as_tensor(user, dtype=torch.int64), torch.as_tensor(item, dtype=torch.int64)
        timestamp = torch.as_tensor(timestamp, dtype=torch.int64).to(user.device)
        rating = torch.as_tensor(rating).to(user.device)
        self.is_train, self.is_val, self.is_test = split_ratings(user, timestamp, train_prop=train_prop, val_prop=val_prop)
        o1 = torch.argsort(timestamp, stable=True)
        self.order = o1[torch.argsort(user[o1], stable=True)]
        su, st = user[self.order], timestamp[self.order]
        self.sorted_user, self.sorted_item, self.sorted_time = su, item[self.order], st
        self.sorted_rating = rating[self.order]
        self.sorted_train = self.is_train[self.order]
        self.sorted_val, self.sorted_test = self.is_val[self.order], self.is_test[self.order]
        self.num_user_rows = int(user.max()) + 1
        # rolling 4-week window of every rating: a slice of the (user, time)-sorted rows.  One composite key orders
        # (user, time) pairs; its offset form turns "t - 4w < t' < t inside the same user" into two searchsorted calls.
        span = int(st.max() - st.min()) + FOUR_WEEKS + 2
        key = su * span + (st - st.min())
        self.history_lo = torch.searchsorted(key, key - FOUR_WEEKS, right=True)        # first t' > t - 4w ...
        seg, start, _ = _segment_layout(su)
        self.history_lo = torch.maximum(self.history_lo, start[seg])                   # ... of the same user
        self.history_hi = torch.searchsorted(key, key, right=False)                    # first t' >= t
        # train pairs and per-user positives
        tr = self.sorted_train
        self.pair_user, self.pair_item = su[tr], self.sorted_item[tr]
        self.pair_target = self.sorted_rating[tr].to(torch.float32)
        counts = torch.bincount(self.pair_user, minlength=self.num_user_rows)
        self.pos_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=user.device), torch.cumsum(counts, 0)])
        self.pos_items = self.pair_item

    def eval_sets(self, split: str):
        """``(users, history_csr, target_csr)`` of ``split`` in {"val", "test"}: ``users`` ascending; ``history_csr`` =
        (offsets, train item rows) to exclude; ``target_csr`` = (offsets, item rows, ratings) to retrieve -- the shapes
        ``MatrixFactorizationLitModule.update_metrics`` consumes."""
        flag = {"val": self.sorted_val, "test": self.sorted_test}[split]
        users = torch.unique(self.sorted_user[flag])
        dev = users.device
        is_eval_user = torch.zeros(self.num_user_rows, dtype=torch.bool, device=dev)
        is_eval_user[users] = True
        slot = torch.full((self.num_user_rows,), -1, dtype=torch.int64, device=dev)
        slot[users] = torch.arange(users.numel(), device=dev)

        def csr(mask):
            rows = slot[self.sorted_user[mask]]                       # already grouped by user, ascending
            off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(torch.bincount(rows, minlength=users.numel()), 0)])
            return off, mask

        h_off, h_mask = csr(self.sorted_train & is_eval_user[self.sorted_user])
        t_off, t_mask = csr((~self.sorted_train) & is_eval_user[self.sorted_user])
        return (users, (h_off, self.sorted_item[h_mask]),
                (t_off, self.sorted_item[t_mask], self.sorted_rating[t_mask].to(torch.float32)))

    def sampler(self, *, num_items: int, batch_size: int = BATCH_SIZE, seed: int = 0, device="cuda", pos_pad: int | None = None,
                user_range: tuple[int, int] | None = None) -> "DeviceInteractionSampler":
        return DeviceInteractionSampler(self.pair_user, self.pair_item, self.pair_target, self.pos_off, self.pos_items,
                                        num_items=num_items, batch_size=batch_size, pos_pad=pos_pad, seed=seed, device=device,
                                        user_range=user_range)


class DeviceInteractionSampler:
    """The training batches, produced on the GPU from HBM-resident interactions (``mf_sample_batch``):
    the counterpart of ``InteractionProcessor.get_batch_data`` (xfmr_rec/data/lightning.py:311-363) for
    id-only towers.  ``pair_user / pair_item / pair_target``: one entry per rating; the positive list
    of user u is ``pos_items[pos_off[u] : pos_off[u + 1]]`` (``UserProcessor.process``, :274-280).
    ``batch(step)`` is example positions ``step * batch_size ...`` of a stream that is reshuffled
    every epoch; the same (seed, step) always gives the same batch.

    Positives.  The reference passes ALL of a user's targets and pads to the longest list of the batch
    (data/lightning.py:275-279, load.py:38-55).  Default (``pos_pad=None``): **no padded tensor at all** -- a batch carries
    ``user.pos_csr = (user idx, pos_off, pos_items)``, the sampler's own HBM-resident lists, and the mask kernels read them
    in place (``mf_loss_fwd_csr``): nothing is ever dropped, and a user with 30,000 ratings costs 30,000 list reads, not a
    30,000-column matrix.  ``pos_pad=P``: the reference's layout, ``user.pos_idx[B, P]`` 0-padded on the right; a P shorter
    than the longest list of the data would silently turn positives into negatives and is refused unless
    ``truncate_positives=True`` says that this is wanted.  ``user_range=(lo, hi)`` keeps only the pairs of users
    ``lo <= u < hi``: the per-rank stream of a user-partitioned job (``distributed.ShardedTrainer(user_mode="partitioned")``)."""

    def __init__(self, pair_user, pair_item, pair_target, pos_off, pos_items, *, num_items: int,
                 batch_size: int = BATCH_SIZE, pos_pad: int | None = None, seed: int = 0, device="cuda",
                 user_range: tuple[int, int] | None = None, truncate_positives: bool = False) -> None:
        from . import _lib

        self._lib = _lib
        i64 = lambda t: torch.as_tensor(t, dtype=torch.int64).to(device).contiguous()  # noqa: E731
        self.pair_user, self.pair_item = i64(pair_user), i64(pair_item)
        self.pair_target = torch.as_tensor(pair_target, dtype=torch.float32).to(device).contiguous()
        if user_range is not None:
            keep = (self.pair_user >= user_range[0]) & (self.pair_user < user_range[1])
            self.pair_user, self.pair_item, self.pair_target = (self.pair_user[keep].contiguous(), self.pair_item[keep].contiguous(),
                                                                self.pair_target[keep].contiguous())
        self.pos_off, self.pos_items = i64(pos_off), i64(pos_items)
        if self.pos_items.numel() == 0:
            self.pos_items = torch.zeros(1, dtype=torch.int64, device=device)
        n = self.pair_user.numel()
        if not (self.pair_item.numel() == n == self.pair_target.numel()) or n == 0:
            msg = "pair_user, pair_item and pair_target must have the same, non-zero length"
            raise ValueError(msg)
        # one-time checks on the data (host syncs at construction, none per batch)
        if int(self.pair_user.min()) < 0 or int(self.pair_user.max()) + 1 >= self.pos_off.numel():
            msg = f"every user id needs a row in pos_off: {int(self.pair_user.max()) = }, {self.pos_off.numel() = }"
            raise ValueError(msg)
        longest = int((self.pos_off[1:] - self.pos_off[:-1]).max()) if self.pos_off.numel() > 1 else 0
        if pos_pad is None:
            pos_pad = 0                      # CSR mode: batches carry the lists themselves
        elif longest > pos_pad and not truncate_positives:
            msg = (f"pos_pad = {pos_pad} would drop positives (longest list: {longest}); the dropped items would enter the "
                   "losses as negatives.  Pass pos_pad=None (sized from the data) or truncate_positives=True")
            raise ValueError(msg)
        self.longest_positive_list = longest
        self.num_items, self.batch_size, self.pos_pad, self.seed = int(num_items), int(batch_size), int(pos_pad), int(seed)

    @property
    def steps_per_epoch(self) -> int:
        return -(-self.pair_user.numel() // self.batch_size)

    def batch(self, step: int) -> InteractionBatchType:
        b, p, dev = self.batch_size, self.pos_pad, self.pair_user.device
        user = torch.empty(b, dtype=torch.int64, device=dev)
        item = torch.empty(2 * b, dtype=torch.int64, device=dev)
        target = torch.empty(b, dtype=torch.float32, device=dev)
        pos = torch.empty(b, p, dtype=torch.int64, device=dev) if p > 0 else None
        lib = self._lib
        lib.check(lib.lib().mf_sample_batch(self.pair_user.data_ptr(), self.pair_item.data_ptr(), self.pair_target.data_ptr(),
                                            self.pair_user.numel(), self.pos_off.data_ptr(), self.pos_items.data_ptr(),
                                            self.num_items, self.seed, int(step) * b, b, p, user.data_ptr(), item.data_ptr(),
                                            target.data_ptr(), None if pos is None else pos.data_ptr(), lib.stream_ptr()))
        ub = {"idx": user, "pos_idx": pos} if pos is not None else {"idx": user, "pos_csr": (user, self.pos_off, self.pos_items)}
        return {"target": target, "user": ub, "item": {"idx": item[:b]}, "neg_item": {"idx": item[b:]}}

    def __iter__(self) -> Iterator[InteractionBatchType]:
        step = 0
        while True:
            yield self.batch(step)
            step += 1


# ------------------------------------------------------------------ MovieLens ratings files (ids only) ---
def _read_id_column(path, sep: str, skip_header: bool):
    """First field of every line of an ids file (movies.dat / users.dat / movies.csv): the reference numbers its rows
    ``movie_rn`` / ``user_rn`` in FILE order from 1 (``with_row_index(..., offset=1)``, prepare.py:85,121)."""
    import numpy as np

    ids = []
    with open(path, encoding="iso-8859-1") as f:
        if skip_header:
            next(f, None)
        for line in f:
            if line.strip():
                ids.append(int(line.split(sep, 1)[0]))
    return np.asarray(ids, dtype=np.int64)


def read_ratings(path) -> dict[str, torch.Tensor]:
    """A MovieLens ratings file as id vectors: ``ml-1m/ratings.dat`` (``UserID::MovieID::Rating::Timestamp``, the file
    ``load_ratings`` reads, xfmr_rec/data/prepare.py:132-152) or ``ml-25m/ratings.csv`` (``userId,movieId,rating,timestamp``
    with a header; half-star ratings).  Returns ``user_id``, ``movie_id``, ``timestamp`` (int64) and ``rating`` (float32)."""
    import io
    import pathlib

    import numpy as np
    import pandas as pd

    path = pathlib.Path(path)
    if path.suffix == ".dat":           # "::" is not a single-character separator: rewrite it and let the C parser run
        raw = path.read_bytes().replace(b"::", b",")
        df = pd.read_csv(io.BytesIO(raw), header=None, names=["user_id", "movie_id", "rating", "timestamp"],
                         dtype={"user_id": np.int64, "movie_id": np.int64, "rating": np.float32, "timestamp": np.int64})
    else:
        df = pd.read_csv(path, header=0, names=["user_id", "movie_id", "rating", "timestamp"],
                         dtype={"user_id": np.int64, "movie_id": np.int64, "rating": np.float32, "timestamp": np.int64})
    return {k: torch.from_numpy(df[k].to_numpy().copy()) for k in ("user_id", "movie_id", "rating", "timestamp")}


def find_movielens(root=None):
    """The first MovieLens ratings file under ``root`` (default: ``$MF_MOVIELENS_DIR``, then ``./data``): the reference's
    ``data/ml-1m/ratings.dat``, or ``ml-25m/ratings.csv`` / ``ml-latest*/ratings.csv``.  ``None`` when there is none
    (there is no network here: benchmarks and fixtures never depend on the files)."""
    import os
    import pathlib

    roots = [root] if root is not None else [os.environ.get("MF_MOVIELENS_DIR"), "data"]
    for r in roots:
        if not r:
            continue
        r = pathlib.Path(r)
        for rel in ("ml-25m/ratings.csv", "ratings.csv", "ml-1m/ratings.dat", "ratings.dat", "ml-20m/ratings.csv",
                    "ml-latest/ratings.csv", "ml-latest-small/ratings.csv", "ml-100k/ratings.csv"):
            if (r / rel).is_file():
                return r / rel
    return None


def movielens_interactions(ratings_path, *, train_prop: float = 0.8, val_prop: float = 0.2):
    """``(InteractionTable, meta)`` of a MovieLens ratings file.  Item rows are the reference's ``movie_rn``: the 1-based
    position of the movie in ``movies.dat`` / ``movies.csv`` next to the ratings file (file order, prepare.py:85); users
    likewise from ``users.dat``; without those files, the rank of the id among the ids that occur (the same numbers whenever
    the side files are sorted by id and complete, as ML-1M's are).  Row 0 of both tables stays the padding row."""
    import pathlib

    import numpy as np

    ratings_path = pathlib.Path(ratings_path)
    r = read_ratings(ratings_path)

    def row_numbers(ids: torch.Tensor, side_files):
        for name, sep, header in side_files:
            f = ratings_path.parent / name
            if f.is_file():
                order = _read_id_column(f, sep, header)
                break
        else:
            order = np.unique(ids.numpy())
        lut = np.full(int(max(order.max(), int(ids.max()))) + 1, -1, dtype=np.int64)
        lut[order] = np.arange(1, order.size + 1)
        rn = lut[ids.numpy()]
        if (rn < 0).any():
            msg = f"{ratings_path}: ids without a row in the side file"
            raise ValueError(msg)
        return torch.from_numpy(rn), int(order.size) + 1

    item, num_items = row_numbers(r["movie_id"], (("movies.dat", "::", False), ("movies.csv", ",", True)))
    user, num_users = row_numbers(r["user_id"], (("users.dat", "::", False),))
    table = InteractionTable(user, item, r["rating"], r["timestamp"], train_prop=train_prop, val_prop=val_prop)
    meta = {"path": str(ratings_path), "num_users": num_users, "num_items": num_items, "ratings": int(user.numel()),
            "train_pairs": int(table.pair_user.numel())}
    return table, meta


def to_device(batch, device):
    if isinstance(batch, dict):
        return {k: to_device(v, device) for k, v in batch.items()}
    if isinstance(batch, (tuple, list)):             # ``pos_csr`` / ``history`` triples
        return type(batch)(to_device(v, device) for v in batch)
    return batch.to(device) if isinstance(batch, torch.Tensor) else batch
