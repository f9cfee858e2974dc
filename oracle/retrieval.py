"""CPU restatement of full-catalog retrieval (TEST INFRASTRUCTURE).

Follows the *semantics* of ``ItemProcessor.search``
(xfmr_rec/data/lightning.py:237-259): cosine score of the query against every
indexed item (embeddings are unit-norm, so score = dot = 1 - cosine distance),
prefilter ``NOT IN (exclude)``, best ``top_k`` first.  The reference delegates the
arithmetic to LanceDB's approximate IVF_HNSW_PQ index (absent here), so the exact
order -- score descending, item row index ascending -- is OUR spec; it is pinned on CPU
against fp64 ``q @ E.T`` + a stable sort on tie-free data and on constructed ties
(tests/test_oracle_pins.py).
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


METRIC_NAMES = ("RetrievalNormalizedDCG", "RetrievalRecall", "RetrievalPrecision", "RetrievalMAP", "RetrievalHitRate",
                "RetrievalMRR")


def retrieval_metrics(topk_idx, targets, k: int) -> np.ndarray:
    """Per-query metrics @k, [Q, 6] in METRIC_NAMES order.  Candidate set and scoring as in
    ``update_metrics`` (xfmr_rec/lightning.py:149-187): retrieved items keep their order, every
    target the search missed is appended below them (the reference draws -U(0,1) for those; any
    order among them gives the same @k values once k items were retrieved); relevance = rating,
    binary relevance = rating > 0 except for NDCG.  Formulas: torchmetrics 1.8 functional
    ``retrieval_normalized_dcg / recall / precision / average_precision / hit_rate /
    reciprocal_rank`` with ``top_k=k`` (torchmetrics is not installed here; pinned by hand-worked cases of those
    formulas -- fewer than k retrieved, rating-0 targets -- in tests/test_oracle_pins.py),
    empty-target queries score 0 (``empty_target_action="neg"``).
    ``targets``: per query a dict item id -> rating."""
    out = np.zeros((len(targets), 6), dtype=np.float64)
    for q, tgt in enumerate(targets):
        got = [int(i) for i in np.asarray(topk_idx[q]).tolist() if int(i) >= 0]
        cand = got + [i for i in tgt if i not in set(got)]
        rel = np.array([float(tgt.get(i, 0.0)) for i in cand])          # already in descending-pred order
        if not (rel > 0).any():
            continue
        top = rel[:k]
        disc = 1.0 / np.log2(np.arange(len(top)) + 2.0)
        ideal = np.sort(rel)[::-1][:k]
        idcg = float((ideal / np.log2(np.arange(len(ideal)) + 2.0)).sum())
        hits = top > 0
        nh = int(hits.sum())
        out[q, 0] = float((top * disc).sum()) / idcg if idcg > 0 else 0.0
        out[q, 1] = nh / int((rel > 0).sum())
        out[q, 2] = nh / k
        if nh:
            pos = np.nonzero(hits)[0] + 1
            out[q, 3] = float(((np.arange(nh) + 1) / pos).mean())
            out[q, 4] = 1.0
            out[q, 5] = 1.0 / pos[0]
    return out
