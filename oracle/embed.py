"""CPU restatement of the embedding-table tower and its row updates
(TEST INFRASTRUCTURE).  The reference's tower is a BERT text
encoder ending in L2-normalisation (xfmr_rec/models.py:42-63) and its optimiser is
dense ``torch.optim.AdamW`` (xfmr_rec/lightning.py:238-239); embedding tables, SGD
and row-wise (lazy) Adam are the north-star's replacement and have no reference
implementation.  Pinned on CPU against third-party definitions
(tests/test_oracle_pins.py): ``gather`` == F.embedding + F.normalize, ``sgd_update`` /
``adam_update`` == torch.optim.SGD / AdamW restricted to the touched rows (duplicates summed,
weight decay, several steps).  **Parity unpinned** for the hash towers' combination rule and
``init_rows`` (our spec).
"""
from __future__ import annotations

import torch


def gather(table: torch.Tensor, idx: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """rows = table[idx]; optionally L2-normalised like models.Normalize() (models.py:59)."""
    rows = table[idx]
    if normalize:
        rows = rows / rows.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return rows


_M64 = (1 << 64) - 1


def hash_buckets(idx: torch.Tensor, num_hashes: int, seed: int, num_buckets: int) -> torch.Tensor:
    """``[*, num_hashes]`` bucket rows: SplitMix64 output function of id + seed + (j + 1) * golden gamma,
    modulo num_buckets (include/mf_numerics.h ``mf_hash_bucket``; our spec).  Python ints: exact."""
    out = []
    for x in idx.reshape(-1).tolist():
        for j in range(num_hashes):
            z = (x + seed + (j + 1) * 0x9E3779B97F4A7C15) & _M64
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
            z ^= z >> 31
            out.append(z % num_buckets)
    return torch.tensor(out, dtype=torch.int64).reshape(*idx.shape, num_hashes)


def gather_hashed(table: torch.Tensor, idx: torch.Tensor, num_hashes: int, seed: int, normalize: bool = True) -> torch.Tensor:
    """Bloom embedding: sum of the id's bucket rows (hash order), optionally L2-normalised."""
    b = hash_buckets(idx, num_hashes, seed, table.shape[0])
    rows = table[b[..., 0]]
    for j in range(1, num_hashes):
        rows = rows + table[b[..., j]]
    if normalize:
        rows = rows / rows.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return rows


def normalize_backward(rows_raw: torch.Tensor, grad_out: torch.Tensor) -> torch.Tensor:
    """d/d raw of (raw / max(||raw||, 1e-12)) applied to grad_out (row-wise)."""
    nrm = rows_raw.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    unit = rows_raw / nrm
    return (grad_out - unit * (grad_out * unit).sum(-1, keepdim=True)) / nrm


def coalesce(idx: torch.Tensor, grad: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Sum gradient rows of duplicate indices (batch order inside a duplicate group)."""
    uniq, inv = torch.unique(idx, return_inverse=True)
    acc = torch.zeros(uniq.numel(), grad.shape[1], dtype=grad.dtype)
    acc.index_add_(0, inv, grad)
    return uniq, acc


def sgd_update(table: torch.Tensor, idx: torch.Tensor, grad: torch.Tensor, lr: float,
               weight_decay: float = 0.0) -> None:
    """In place: table[r] -= lr * (sum of grads of r + wd * table[r]) for touched rows."""
    uniq, acc = coalesce(idx, grad)
    rows = table[uniq]
    table[uniq] = rows - lr * (acc + weight_decay * rows)


def adam_update(table, m, v, idx, grad, *, step: int, lr: float, beta1: float = 0.9,
                beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0) -> None:
    """Row-wise lazy AdamW: only touched rows move; moments decay only when touched;
    bias correction uses the global ``step`` (1-based); decoupled weight decay as in
    torch.optim.AdamW (the reference's optimiser class, lightning.py:238-239)."""
    uniq, g = coalesce(idx, grad)
    p = table[uniq] * (1.0 - lr * weight_decay)
    mm = m[uniq] * beta1 + (1.0 - beta1) * g
    vv = v[uniq] * beta2 + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (vv / bc2).sqrt() + eps
    table[uniq] = p - (lr / bc1) * mm / denom
    m[uniq] = mm
    v[uniq] = vv


def init_rows(n_local: int, d: int, row_start: int, row_stride: int, seed: int, std: float) -> torch.Tensor:
    """Shard initialisation of the sharded path (``mf_init_rows``, csrc/mf_comm.hip; our spec): local row l is global row
    ``row_start + l * row_stride`` of a virtual table whose element (row, col) is ``std`` times a normal variate that is a
    pure function of (seed, row, col): SplitMix64 of ``seed + (row * d + col) * golden gamma``, two 24-bit uniforms,
    Box-Muller.  numpy restatement (uint64 wrap-around arithmetic)."""
    import numpy as np

    rows = np.uint64(row_start) + np.arange(n_local, dtype=np.uint64) * np.uint64(row_stride)
    idx = rows[:, None] * np.uint64(d) + np.arange(d, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u1 = ((z >> np.uint64(40)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    u2 = (((z >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    out = np.float32(std) * np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
    return torch.from_numpy(out.astype(np.float32))


def logq_from_counts(counts: torch.Tensor) -> torch.Tensor:
    """log of the empirical sampling probability of each item (Yi et al. 2019; the
    README cites it, README.md:29-30, but the reference never implements it)."""
    p = counts.to(torch.float64).clamp_min(1.0)
    return (p / p.sum()).log().to(torch.float32)
