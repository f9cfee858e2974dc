"""Exact brute-force retrieval over the item table (replaces the LanceDB ANN index).

Reference: ``ItemProcessor.get_index`` embeds every item and builds an
``IVF_HNSW_PQ`` cosine index (xfmr_rec/data/lightning.py:182-235);
``ItemProcessor.search`` runs one query with a ``movie_id NOT IN (...)`` prefilter
and returns the ``top_k`` rows with ``score = 1 - cosine distance``, best first
(:237-259).  Here the "index" is the unit-norm item matrix itself, resident in HBM,
and ``search`` is an exact scan (``mf_topk``): same scores for unit-norm rows, exact
instead of approximate, batched over queries, ties broken by lowest item row.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import _lib
from .params import ITEM_ID_COL, ITEM_IDX_COL, TOP_K


def _csr(exclude: Sequence[Sequence[int]] | None, q: int, device):
    """Per-query exclusion lists -> (offsets[Q+1], sorted row indices) on the GPU."""
    if exclude is None:
        return None, None
    if len(exclude) != q:
        msg = f"one exclusion list per query expected: {len(exclude) = }, {q = }"
        raise ValueError(msg)
    lens = np.fromiter((len(e) for e in exclude), dtype=np.int64, count=q)
    off = np.zeros(q + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.concatenate([np.sort(np.asarray(e, dtype=np.int64)) for e in exclude]) if off[-1] else np.zeros(1, np.int64)
    return torch.from_numpy(off).to(device), torch.from_numpy(flat).to(device)


class ItemIndex:
    """Item embeddings ``[N, d]`` (row r = global item row ``idx_base + r``) on one GPU.  A snapshot, like the table the
    reference's ``get_index`` writes: the derived copies (``blocked()``, ``bf16_index()``) are built on first use from the
    rows as they are then -- ``refresh()`` after changing the rows in place."""

    def __init__(self, embeddings: torch.Tensor, *, idx_base: int = 0) -> None:
        emb = _lib.dev_f32(embeddings.detach(), "embeddings")
        self.dim = emb.shape[1]
        dp = _lib.padded_width(self.dim)
        if dp != self.dim:
            emb = torch.nn.functional.pad(emb, (0, dp - self.dim))
        self.embeddings = emb.contiguous()
        self.idx_base = int(idx_base)
        self._blocked: torch.Tensor | None = None      # second copy for the small-batch scan, built on first use
        self._bf16: torch.Tensor | None = None         # bf16 copy + largest row norm for the prefilter path, built on first use
        self._ws: dict = {}                              # search workspaces, kept between calls

    SMALL_Q = 32       # at most this many queries can take the bandwidth-bound matrix-vector path (mf_topk_small)
    # "auto" (measured at N = 62,423, d = 128, us of device time per call): scan 15 at Q = 1 (FMA chains, HBM-bound), and for
    # 2 .. 32 queries -- one MFMA tile side: the fp32 matrix core computes all of them in the time of one -- see
    # profiles/r03_topk_small_probe.log; bf16 prefilter 42 / 44 / 51 / 70 at Q = 32 / 256 / 512 / 1024 (five launches: ~35 us floor);
    # fp32 tiles 95 / 86 / 98 / 133 / 219
    AUTO_SMALL_Q = 32  # scan up to here, then the bf16 prefilter (d >= 64), else the fp32 tile engine
    SCAN_MAX_EXCL = 8192     # exclusion entries (all queries of a call) the scan stages in LDS: EX_CAP of mf_topk_small.hip

    def blocked(self) -> torch.Tensor:
        """The catalog in the blocked layout of ``mf_topk_small`` (``[64-row block][chunk][row]``), built once."""
        if self._blocked is None:
            lib = _lib.lib()
            n, d = self.embeddings.shape
            out = torch.empty(lib.mf_topk_blocked_bytes(n, d) // 4, dtype=torch.float32, device=self.embeddings.device)
            _lib.check(lib.mf_topk_blocked_build(self.embeddings.data_ptr(), n, d, out.data_ptr(), _lib.stream_ptr()))
            self._blocked = out
        return self._blocked

    BF16_MIN_Q = 5     # "auto" takes the bf16-prefilter path from here (and d >= 64, N >= BF16_MIN_N)
    BF16_MIN_N = 2048

    def bf16_index(self) -> torch.Tensor:
        """bf16 rows + the largest row norm, the prefilter index of ``mf_topk_bf3``; built once."""
        if self._bf16 is None:
            lib = _lib.lib()
            n, d = self.embeddings.shape
            out = torch.empty(lib.mf_topk_bf3_index_bytes(n, d), dtype=torch.uint8, device=self.embeddings.device)
            _lib.check(lib.mf_topk_bf3_build(self.embeddings.data_ptr(), n, d, out.data_ptr(), out.numel(), _lib.stream_ptr()))
            self._bf16 = out
        return self._bf16

    def refresh(self) -> None:
        """Forget the derived copies: they are rebuilt from the current rows at the next search."""
        self._blocked = self._bf16 = None

    def _workspace(self, key, nbytes: int) -> torch.Tensor:
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[key] = _lib.workspace(nbytes, self.embeddings.device)
        return ws

    @property
    def num_items(self) -> int:
        return self.embeddings.shape[0]

    def search(self, queries: torch.Tensor, top_k: int = TOP_K, *, exclude: Sequence[Sequence[int]] | None = None,
               exclude_csr: tuple[torch.Tensor, torch.Tensor] | None = None,
               path: str = "auto") -> tuple[torch.Tensor, torch.Tensor]:
        """``(scores [Q, k] fp32, rows [Q, k] int64)``, best first; ``exclude`` holds GLOBAL
        item rows per query (or pass a prebuilt device CSR with sorted ids).  ``path``: "scan" = the
        matrix-vector kernel (at most ``SMALL_Q`` queries), "tiles" = the fp32 MFMA tile engine, "bf16" = two bf16
        MFMA scans that pick the few dozen rows per query worth an exact fp32 score (d >= 64), "auto" picks by the
        number of queries; all three give the same bits."""
        q = _lib.dev_f32(queries, "queries")
        if q.dim() != 2 or q.shape[1] != self.dim:
            msg = f"queries should be (num_queries, {self.dim}): {tuple(q.shape) = }"
            raise ValueError(msg)
        if self.embeddings.shape[1] != self.dim:
            q = torch.nn.functional.pad(q, (0, self.embeddings.shape[1] - self.dim))
        nq, n, d = q.shape[0], self.num_items, self.embeddings.shape[1]
        off, ids = exclude_csr if exclude_csr is not None else _csr(exclude, nq, q.device)
        lib = _lib.lib()
        scores = torch.empty(nq, top_k, dtype=torch.float32, device=q.device)
        rows = torch.empty(nq, top_k, dtype=torch.int64, device=q.device)
        if path not in ("auto", "scan", "tiles", "bf16"):
            msg = f"path must be 'auto', 'scan', 'tiles' or 'bf16': {path = }"
            raise ValueError(msg)
        if path == "bf16" and d < 64:
            msg = f"the bf16 path needs an embedding width of at least 64: {d = }"
            raise ValueError(msg)
        if path == "scan" and nq > self.SMALL_Q:
            msg = f"the scan path takes at most {self.SMALL_Q} queries: {nq = }"
            raise ValueError(msg)
        # the few-query scan matches exclusion entries against every 64-row block: staged in LDS up to SCAN_MAX_EXCL entries
        # (all queries together), a walk over each query's list per block beyond -- measured at N = 62,423: Q = 8 x 2,000 ids
        # 89 us, Q = 32 x 30,000 ids 4 ms, against 46 / 85 us through the prefilter (which builds bit rows once).  So "auto"
        # sends long lists -- a heavy user's history -- to the other engines
        entries = 0 if ids is None else int(ids.numel())
        bf16_ok = n >= self.BF16_MIN_N and d >= 64
        scan_ok = nq <= self.AUTO_SMALL_Q and (entries <= self.SCAN_MAX_EXCL or not bf16_ok and entries <= 16 * self.SCAN_MAX_EXCL)
        if path == "scan" or (path == "auto" and scan_ok):
            # the reference's own shape: one query (or a handful) per call -- bandwidth-bound scan, two launches
            ws = self._workspace(("small", nq, top_k), lib.mf_topk_small_ws_bytes(nq, n, d, top_k))
            _lib.check(lib.mf_topk_small(q.data_ptr(), nq, self.blocked().data_ptr(), n, d, top_k, _lib.ptr(off), _lib.ptr(ids),
                                         self.idx_base, ws.data_ptr(), ws.numel(), scores.data_ptr(), rows.data_ptr(),
                                         _lib.stream_ptr()))
            return scores, rows
        if path == "bf16" or (path == "auto" and (nq >= self.BF16_MIN_Q or not scan_ok) and bf16_ok):
            ws = self._workspace(("bf16", nq, top_k), lib.mf_topk_bf3_ws_bytes(nq, n, d, top_k))
            rc = lib.mf_topk_bf3(q.data_ptr(), nq, self.embeddings.data_ptr(), self.bf16_index().data_ptr(), n, d, top_k,
                                 _lib.ptr(off), _lib.ptr(ids), self.idx_base, ws.data_ptr(), ws.numel(), scores.data_ptr(),
                                 rows.data_ptr(), _lib.stream_ptr())
            # a shape outside the prefilter's descriptor limits (MF_ENOTSUP, decided on the host before any launch) goes
            # to the fp32 tile engine when the caller left the choice to us; an explicit path="bf16" raises
            if not (rc == _lib.MF_ENOTSUP and path == "auto"):
                _lib.check(rc)
                return scores, rows
        ws = self._workspace(("tile", nq, top_k), lib.mf_topk_ws_bytes(nq, n, d, top_k))
        _lib.check(lib.mf_topk(q.data_ptr(), nq, self.embeddings.data_ptr(), n, d, top_k, _lib.ptr(off), _lib.ptr(ids),
                               self.idx_base, ws.data_ptr(), ws.numel(), scores.data_ptr(), rows.data_ptr(),
                               _lib.stream_ptr()))
        return scores, rows


def merge_topk(part_scores: torch.Tensor, part_rows: torch.Tensor, top_k: int) -> tuple[torch.Tensor, torch.Tensor]:
    """Merge per-shard results ``[G, Q, k]`` (global rows) into the global top-k (``mf_topk_merge``)."""
    ps = _lib.dev_f32(part_scores, "part_scores")
    pr = _lib.dev_i64(part_rows, "part_rows")
    g, nq, k = ps.shape
    if k != top_k:
        msg = f"partial results must hold top_k entries: {k = }, {top_k = }"
        raise ValueError(msg)
    scores = torch.empty(nq, k, dtype=torch.float32, device=ps.device)
    rows = torch.empty(nq, k, dtype=torch.int64, device=ps.device)
    _lib.check(_lib.lib().mf_topk_merge(ps.data_ptr(), pr.data_ptr(), g, nq, k, scores.data_ptr(), rows.data_ptr(),
                                        _lib.stream_ptr()))
    return scores, rows


class ItemProcessor:
    """The retrieval half of ``xfmr_rec.data.lightning.ItemProcessor`` (:154-259): ``get_index``
    embeds the catalog with the item tower, ``search`` answers one query like the reference
    (numpy ``[1, d]`` in, ``pandas.DataFrame`` out with ``movie_rn``, ``movie_id``, ``score``)."""

    idx_col: str = ITEM_IDX_COL
    id_col: str = ITEM_ID_COL

    def __init__(self, item_ids: Sequence[int] | torch.Tensor | None = None) -> None:
        self.item_ids = None if item_ids is None else torch.as_tensor(item_ids, dtype=torch.int64).cpu()
        self.index: ItemIndex | None = None
        self._row_of_id: dict[int, int] | None = None

    @torch.inference_mode()
    def get_index(self, model: torch.nn.Module, subset: str = "predict") -> ItemIndex:  # noqa: ARG002
        tower = model.towers["item"] if hasattr(model, "towers") else model
        rows = torch.arange(tower.num_embeddings, device=tower.weight.device)
        self.index = ItemIndex(tower(rows))
        if self.item_ids is None:
            self.item_ids = torch.arange(tower.num_embeddings)
        self._row_of_id = {int(i): r for r, i in enumerate(self.item_ids.tolist())}
        return self.index

    def set_index(self, embeddings: torch.Tensor) -> ItemIndex:
        """Install a prebuilt (loaded) item matrix; row r belongs to ``item_ids[r]``."""
        self.index = ItemIndex(embeddings)
        if self.item_ids is None:
            self.item_ids = torch.arange(embeddings.shape[0])
        self._row_of_id = {int(i): r for r, i in enumerate(self.item_ids.tolist())}
        return self.index

    def row_of(self, item_id: int) -> int:
        if self._row_of_id is None or int(item_id) not in self._row_of_id:
            msg = f"unknown item id: {item_id = }"
            raise KeyError(msg)
        return self._row_of_id[int(item_id)]

    def search(self, embedding, exclude_item_ids: list[int] | None = None, top_k: int = TOP_K):
        import pandas as pd

        if self.index is None:
            msg = "`index` must be intialised first"  # message kept from data/lightning.py:244
            raise ValueError(msg)
        q = torch.as_tensor(np.asarray(embedding), dtype=torch.float32).reshape(1, -1).to(self.index.embeddings.device)
        rows = [self._row_of_id[i] for i in (exclude_item_ids or []) if i in self._row_of_id]
        scores, idx = self.index.search(q, top_k, exclude=[rows])
        idx = idx[0].cpu()
        keep = idx >= 0
        idx = idx[keep]
        return pd.DataFrame(
            {
                self.idx_col: idx.numpy(),
                self.id_col: self.item_ids[idx].numpy(),
                "score": scores[0].cpu()[keep].numpy(),
            }
        )


METRIC_NAMES = ("RetrievalNormalizedDCG", "RetrievalRecall", "RetrievalPrecision", "RetrievalMAP", "RetrievalHitRate",
                "RetrievalMRR")


class RetrievalMetrics:
    """The reference's ``MetricCollection`` of six retrieval metrics @top_k
    (xfmr_rec/lightning.py:289-306), fed with whole batches of top-k results instead of one
    ``update`` per example (``update_metrics`` :149-187).  ``compute()`` returns
    ``{prefix + class name: mean over all queries seen}`` like ``MetricCollection.compute``."""

    def __init__(self, top_k: int = TOP_K, prefix: str = "") -> None:
        self.top_k, self.prefix = int(top_k), prefix
        self.reset()

    def reset(self) -> None:
        self._sum, self._n = None, 0

    @torch.no_grad()
    def update(self, topk_idx: torch.Tensor, target_offsets: torch.Tensor, target_idx: torch.Tensor,
               target_rating: torch.Tensor) -> torch.Tensor:
        """``topk_idx`` [Q, top_k] retrieved item ids, best first (``ItemIndex.search``); the targets of query q
        are ``target_idx / target_rating[target_offsets[q] : target_offsets[q + 1]]``.  Returns [Q, 6]."""
        ti = _lib.dev_i64(topk_idx, "topk_idx")
        q, k = ti.shape
        if k != self.top_k:
            msg = f"expected top_k = {self.top_k} columns: {k = }"
            raise ValueError(msg)
        off = _lib.dev_i64(target_offsets, "target_offsets")
        ids = _lib.dev_i64(target_idx, "target_idx")
        rel = _lib.dev_f32(target_rating, "target_rating")
        if off.numel() != q + 1 or ids.numel() != rel.numel():
            msg = f"target CSR does not match the queries: {off.numel() = }, {q = }, {ids.numel() = }, {rel.numel() = }"
            raise ValueError(msg)
        if ids.numel() == 0:       # keep the pointers valid
            ids, rel = torch.zeros(1, dtype=torch.int64, device=ti.device), torch.zeros(1, device=ti.device)
        out = torch.empty(q, 6, dtype=torch.float32, device=ti.device)
        _lib.check(_lib.lib().mf_retrieval_metrics(ti.data_ptr(), q, k, off.data_ptr(), ids.data_ptr(), rel.data_ptr(),
                                                   out.data_ptr(), _lib.stream_ptr()))
        tot = out.sum(dim=0, dtype=torch.float64)
        self._sum = tot if self._sum is None else self._sum + tot
        self._n += q
        return out

    def compute(self) -> dict[str, torch.Tensor]:
        if not self._n:
            return {self.prefix + name: torch.tensor(0.0) for name in METRIC_NAMES}
        mean = (self._sum / self._n).to(torch.float32)
        return {self.prefix + name: mean[j] for j, name in enumerate(METRIC_NAMES)}
