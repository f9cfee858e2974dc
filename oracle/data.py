"""CPU restatement of the on-device batch producer (TEST INFRASTRUCTURE; spec of
``mf_sample_batch``, csrc/mf_data.hip).  **Parity unpinned**: the reference assembles batches with a
host datapipe (xfmr_rec/data/lightning.py:311-363) whose shuffle order is torch's; only the batch
LAYOUT is the reference's (InteractionBatchType, data/lightning.py:72-76; 0-padding of ``pos_idx``,
data/load.py:38-55).  Exact integer arithmetic in Python ints.
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
