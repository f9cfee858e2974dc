"""CPU restatement of full-catalog retrieval (TEST INFRASTRUCTURE).

Follows the *semantics* of ``ItemProcessor.search``
(xfmr_rec/data/lightning.py:237-259): cosine score of the query against every
indexed item (embeddings are unit-norm, so score = dot = 1 - cosine distance),
prefilter ``NOT IN (exclude)``, best ``top_k`` first.  The reference delegates the
arithmetic to LanceDB's approximate IVF_HNSW_PQ index (absent here), so the exact
order -- score descending, item row index ascending -- is OUR spec:
**parity unpinned**.
"""
from __future__ import annotations

import numpy as np
import torch

from . import chain


def topk_exact(q, items, k: int, exclude=None):
    """Bit-exact oracle (chain scores, oracle/chain.c).  Slow: O(Q*N*d) scalar."""
    return chain.topk(np.asarray(q), np.asarray(items), k, exclude)


def topk_fast(q: torch.Tensor, items: torch.Tensor, k: int, exclude=None):
    """All-core PyTorch baseline (``q @ items.T`` + topk); used as the timed CPU
    baseline.  Its BLAS summation order differs from the chain, so it is NOT the
    bit-exact checker."""
    s = q @ items.T
    if exclude is not None:
        for r, ex in enumerate(exclude):
            if len(ex):
                s[r, torch.as_tensor(ex, dtype=torch.long)] = float("-inf")
    val, idx = torch.topk(s, k, dim=1)
    return val, idx


def merge_topk(scores: np.ndarray, idx: np.ndarray, k: int):
    """Merge per-shard partial results [G, Q, k] (global item indices) into the
    global top-k with the same (score desc, index asc) order."""
    G, Q, kk = scores.shape
    s = np.transpose(scores, (1, 0, 2)).reshape(Q, G * kk)
    i = np.transpose(idx, (1, 0, 2)).reshape(Q, G * kk)
    out_s = np.empty((Q, k), np.float32)
    out_i = np.empty((Q, k), np.int64)
    for r in range(Q):
        valid = i[r] >= 0
        order = np.lexsort((i[r][valid], -s[r][valid].astype(np.float64)))[:k]
        n = order.size
        out_s[r, :n], out_i[r, :n] = s[r][valid][order], i[r][valid][order]
        out_s[r, n:], out_i[r, n:] = -np.inf, -1
    return out_s, out_i
