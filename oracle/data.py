"""CPU restatement of the on-device batch producer (TEST INFRASTRUCTURE; spec of
``mf_sample_batch``, csrc/mf_data.hip).  **Parity unpinned**: the reference assembles batches with a
host datapipe (xfmr_rec/data/lightning.py:311-363) whose shuffle order is torch's; only the batch
LAYOUT is the reference's (InteractionBatchType, data/lightning.py:72-76; 0-padding of ``pos_idx``,
data/load.py:38-55).  Exact integer arithmetic in Python ints.  ``split_ratings`` / ``rolling_history`` restate the
polars expressions of data/prepare.py:160-194 / :229-243 on ids only and ARE pinned: tests/test_host_cpu.py checks them
against a plain-Python restatement of the same expressions (rank("min") semantics, 4-week window).
"""
from __future__ import annotations

import torch

_M64 = (1 << 64) - 1
_GAMMA = 0x9E3779B97F4A7C15


def splitmix64(z: int) -> int:
    z &= _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def feistel_perm(x: int, n: int, epoch: int, seed: int) -> int:
    """Position x of epoch ``epoch`` -> interaction index: a bijection of [0, n) (cycle-walking)."""
    bits = 1
    while (1 << bits) < n:
        bits += 1
    hb = (bits + 1) // 2
    mask = (1 << hb) - 1
    while True:
        left, right = x >> hb, x & mask
        for rnd in range(4):
            f = splitmix64(right + (rnd << 56) + epoch * _GAMMA + seed) & mask
            left, right = right, left ^ f
        x = (left << hb) | right
        if x < n:
            return x


def sample_batch(pair_user, pair_item, pair_target, pos_off, pos_items, num_items: int, seed: int, start: int,
                 batch: int, pad: int):
    n = len(pair_user)
    user, item, neg, target = [], [], [], []
    pos = torch.zeros(batch, pad, dtype=torch.int64)
    for r in range(batch):
        p = start + r
        e = feistel_perm(p % n, n, p // n, seed)
        u = int(pair_user[e])
        user.append(u)
        item.append(int(pair_item[e]))
        target.append(float(pair_target[e]))
        neg.append(1 + splitmix64(p * 0xD1342543DE82EF95 + seed + 0x632BE59BD9B4E019) % (num_items - 1))
        own = pos_items[int(pos_off[u]): int(pos_off[u + 1])][:pad]
        pos[r, : len(own)] = torch.as_tensor(own, dtype=torch.int64)
    return (torch.tensor(user), torch.tensor(item + neg), torch.tensor(target, dtype=torch.float32), pos)


# ---- ids-only ETL (spec of data.split_ratings / data.InteractionTable): plain Python following the polars expressions of
# ---- xfmr_rec/data/prepare.py:160-194 (train_test_split) and :229-243 (gather_history) line by line
def _rank_min(values):
    """polars rank("min"): 1 + number of strictly smaller values."""
    return [1 + sum(1 for w in values if w < v) for v in values]


def split_ratings(user, timestamp, train_prop: float = 0.8, val_prop: float = 0.2):
    n = len(user)
    by_user: dict[int, list[int]] = {}
    for r in range(n):
        by_user.setdefault(int(user[r]), []).append(r)
    is_train = [False] * n
    for rows in by_user.values():
        ranks = _rank_min([int(timestamp[r]) for r in rows])
        for r, rk in zip(rows, ranks):
            is_train[r] = (rk - 1) / len(rows) < train_prop                      # prepare.py:170-176
    nontrain = {u: sum(1 for r in rows if not is_train[r]) for u, rows in by_user.items()}
    nontrain = {u: c for u, c in nontrain.items() if c > 0}                      # filter(~is_train).group_by(user).len()
    us = list(nontrain)
    ranks = _rank_min([nontrain[u] for u in us])
    val_user = {u: (rk - 1) / len(us) >= 1 - val_prop for u, rk in zip(us, ranks)}   # :181-184
    is_val = [(not is_train[r]) and val_user.get(int(user[r]), False) for r in range(n)]
    is_test = [(not is_train[r]) and not is_val[r] for r in range(n)]
    return is_train, is_val, is_test


def rolling_history(user, item, timestamp, period: int):
    """history of rating r: items of the same user with t - period < t' < t (closed="none"), in time order."""
    n = len(user)
    out = []
    for r in range(n):
        rows = [q for q in range(n) if int(user[q]) == int(user[r]) and int(timestamp[r]) - period < int(timestamp[q]) < int(timestamp[r])]
        rows.sort(key=lambda q: (int(timestamp[q]), q))
        out.append([int(item[q]) for q in rows])
    return out
