"""Multi-GPU hot path: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL
over xGMI on ROCm; ``gloo`` in the CPU tests).

The reference has no explicit collective anywhere (implicit DDP of a
dense BERT); the sharding below is the north-star's and our design:

* **item table row-sharded, rows dealt round-robin** (rank r owns rows ``r, r + G, r + 2G, ...``):
  item popularity is heavy-tailed and id order often follows it, so contiguous blocks would send most
  of every batch to one owner (its gather, sort and update then set the step time of the whole job);
  **user table sharded in contiguous blocks** and the training pairs partitioned by user shard, so
  user rows never travel;
* **training step**: each rank needs the item rows of its own batch (B positives + B
  sampled negatives).  Rows are fetched from their owners with one all-to-all of ids
  and one all-to-all of rows (8 MB per rank at B = 8192, d = 128 -- latency-bound on
  xGMI, so direct all-to-all, never a ring all-reduce of a dense table), the score /
  loss kernels run on the local B x 2B block (in-batch negatives stay local, exactly
  like the reference, which has no cross-rank gather), item-row gradients go back to
  their owners with one more all-to-all and each owner applies ONE sparse update per
  step (duplicates summed in (rank, batch) order: deterministic);
* **retrieval**: queries are all-gathered, every rank scans ITS shard for all queries
  (``mf_topk`` on local rows, mapped to global rows before they leave), the partial top-k travel back to the query's rank by
  all-to-all and are merged exactly (``mf_topk_merge``): bit-identical to the
  single-GPU result.

Local compute goes through an ``ops`` object (default :class:`HipOps`) so that the
communication logic can be exercised on CPU with ``gloo`` by injecting the oracle.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib


class HipOps:
    """Local compute on the GPU through libmf_hip.so (the product path)."""

    def __init__(self, mf) -> None:
        self.mf = mf

    def gather(self, table, ids, normalize):
        out = torch.empty(ids.numel(), table.shape[1], dtype=torch.float32, device=table.device)
        _lib.check(_lib.lib().mf_gather_rows(table.data_ptr(), table.shape[0], table.shape[1], ids.data_ptr(), ids.numel(),
                                             int(normalize), out.data_ptr(), None, _lib.stream_ptr()))
        return out

    def loss_and_grads(self, kind, u, v, target, item_idx, pos_idx, logq, num_negatives, sigma, margin):
        u = u.detach().requires_grad_()
        v = v.detach().requires_grad_()
        fn = getattr(self.mf.losses, kind)(num_negatives=num_negatives, sigma=sigma, margin=margin)
        loss = fn(u, v, target, item_idx=item_idx, pos_idx=pos_idx, logq=logq)
        loss.backward()
        return loss.detach(), u.grad, v.grad

    def update(self, optimizer, table, state, ids, grad, normalized, step, lr):
        lib = _lib.lib()
        n, d = ids.numel(), table.shape[1]
        if n == 0:
            return
        ws = _lib.workspace(lib.mf_update_ws_bytes(n, d), table.device)
        if optimizer == "sgd":
            _lib.check(lib.mf_update_sgd(table.data_ptr(), table.shape[0], d, ids.data_ptr(), n, grad.data_ptr(),
                                         int(normalized), lr, 0.0, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        else:
            _lib.check(lib.mf_update_adam(table.data_ptr(), state["m"].data_ptr(), state["v"].data_ptr(), table.shape[0], d,
                                          ids.data_ptr(), n, grad.data_ptr(), int(normalized), step, None, lr, 0.9, 0.999, 1e-8,
                                          0.01, ws.data_ptr(), ws.numel(), _lib.stream_ptr()))

    def topk(self, queries, items, k, exclude_csr, idx_base):
        return self.mf.retrieval.ItemIndex(items, idx_base=idx_base).search(queries, k, exclude_csr=exclude_csr)

    def merge(self, part_scores, part_rows, k):
        return self.mf.retrieval.merge_topk(part_scores, part_rows, k)


def shard_bounds(n_rows: int, world: int, rank: int) -> tuple[int, int]:
    per = (n_rows + world - 1) // world
    return min(rank * per, n_rows), min((rank + 1) * per, n_rows)


def _all_to_all_rows(x: torch.Tensor, send_counts: list[int], recv_counts: list[int]) -> torch.Tensor:
    out = torch.empty((sum(recv_counts),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_to_all_single(out, x.contiguous(), output_split_sizes=recv_counts, input_split_sizes=send_counts)
    return out


def cyclic_rows(n_rows: int, world: int, rank: int) -> int:
    """Rows of a table dealt round-robin that land on ``rank``."""
    return (n_rows - rank + world - 1) // world if n_rows > rank else 0


class RowExchange:
    """Routes a list of global row ids to their owning ranks and back (rows dealt round-robin:
    owner = id mod world, local row = id div world)."""

    def __init__(self, ids: torch.Tensor, n_rows: int) -> None:  # noqa: ARG002
        world = dist.get_world_size()
        owner = torch.remainder(ids, world)
        self.order = torch.argsort(owner, stable=True)              # batch position of every sent slot
        send = torch.bincount(owner, minlength=world)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send)
        # NCCL/RCCL needs the split sizes on the host: the one host sync of the step
        self.send_counts, self.recv_counts = send.tolist(), recv.tolist()
        # rows of MY shard that the others (and I) asked for, grouped by requesting rank
        self.local_ids = torch.div(_all_to_all_rows(ids[self.order], self.send_counts, self.recv_counts), world,
                                   rounding_mode="floor")

    def fetch(self, rows_for_requests: torch.Tensor) -> torch.Tensor:
        """owner -> requester: rows gathered for ``local_ids`` come back in batch order."""
        got = _all_to_all_rows(rows_for_requests, self.recv_counts, self.send_counts)
        out = torch.empty_like(got)
        out[self.order] = got
        return out

    def push(self, per_batch_rows: torch.Tensor) -> torch.Tensor:
        """requester -> owner: one row per batch slot, delivered aligned with ``local_ids``."""
        return _all_to_all_rows(per_batch_rows[self.order], self.send_counts, self.recv_counts)


class ShardedTrainer:
    """Row-sharded tables + the training step described in the module docstring."""

    def __init__(self, mf, device, optimizer: str, num_negatives: int, *, num_users: int, num_items: int, dim: int,
                 logq: torch.Tensor | None = None, kind: str = "InfomationNoiseContrastiveEstimationLoss",
                 lr: float | None = None, ops=None, seed: int = 0) -> None:
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.ops = ops if ops is not None else HipOps(mf)
        self.optimizer, self.num_negatives, self.kind = optimizer, num_negatives, kind
        self.lr = lr if lr is not None else (1e-4 if optimizer == "adam" else 1e-2)
        self.num_users, self.num_items, self.dim = num_users, num_items, dim
        self.user_lo, self.user_hi = shard_bounds(num_users, self.world, self.rank)
        g = torch.Generator().manual_seed(seed)                     # same full tables on every rank, then sliced
        full_u = torch.randn(num_users, dim, generator=g) / dim**0.5
        full_i = torch.randn(num_items, dim, generator=g) / dim**0.5
        self.user_table = full_u[self.user_lo:self.user_hi].contiguous().to(device)
        self.item_table = full_i[self.rank::self.world].contiguous().to(device)     # rows rank, rank + world, ...
        self.state = {name: {"m": torch.zeros_like(t), "v": torch.zeros_like(t)}
                      for name, t in (("user", self.user_table), ("item", self.item_table))}
        self.logq = logq
        self.steps = 0

    # -- bench helpers ------------------------------------------------------------------------
    def item_shard(self) -> torch.Tensor:
        ids = torch.arange(self.item_table.shape[0], device=self.item_table.device)
        return self.ops.gather(self.item_table, ids, True)

    def item_offset(self) -> int:
        """Global row of local item row l is ``l * item_stride() + item_offset()``."""
        return self.rank

    def item_stride(self) -> int:
        return self.world

    def item_matrix(self) -> torch.Tensor:
        return self.item_shard()

    def user_vectors(self, rows: torch.Tensor) -> torch.Tensor:
        """Unit-norm vectors of global user rows that this rank owns (others wrap into its shard)."""
        local = (rows - self.user_lo) % max(self.user_table.shape[0], 1)
        return self.ops.gather(self.user_table, local, True)

    # -- exchange plans ------------------------------------------------------------------------
    def _plan(self, item_ids: torch.Tensor) -> RowExchange:
        """The routing plan of a batch's item ids.  Building one costs a host sync (the split sizes);
        ``prefetch`` moves that sync off the critical path."""
        hit = self._plans.pop(item_ids.data_ptr(), None) if hasattr(self, "_plans") else None
        if hit is None:
            return RowExchange(item_ids, self.num_items)
        ex, ready = hit
        if ready is not None:                      # built on the side stream: order it before our use
            cur = torch.cuda.current_stream()
            cur.wait_event(ready)
            for t in (ex.order, ex.local_ids):
                t.record_stream(cur)
        return ex

    def prefetch(self, next_b) -> None:
        """Build the plan of the NEXT batch now, on a side stream: its small kernels, its two tiny
        all-to-alls and the host read of the split sizes run beside the current step's MFMA sweeps
        instead of stalling the start of the next step (ids of the next batch are known: prefetching
        loader).  Every rank must call it at the same point of its step (collective order)."""
        if not hasattr(self, "_plans"):
            self._plans, self._plan_stream = {}, None
        ids = next_b["item"]
        if ids.data_ptr() in self._plans:
            return
        if not ids.is_cuda:
            self._plans[ids.data_ptr()] = (RowExchange(ids, self.num_items), None)
            return
        if self._plan_stream is None:
            # high priority: its own hardware queue, so the small plan kernels are dispatched beside the sweeps
            self._plan_stream = torch.cuda.Stream(device=ids.device, priority=-1)
        with torch.cuda.stream(self._plan_stream):
            ex = RowExchange(ids, self.num_items)
            ready = torch.cuda.Event()
            ready.record(self._plan_stream)
        self._plans[ids.data_ptr()] = (ex, ready)

    # -- one step -----------------------------------------------------------------------------
    def step(self, b, next_b=None) -> torch.Tensor:
        """``b``: ``user`` (global rows inside this rank's user shard), ``item`` (2B global rows:
        positives then negatives), ``target``, ``pos``.  ``next_b``: the batch after it, if known
        (its exchange plan is then prefetched behind this step's compute)."""
        self.steps += 1
        user_local = b["user"] - self.user_lo
        ex = self._plan(b["item"])
        v = ex.fetch(self.ops.gather(self.item_table, ex.local_ids, True))
        u = self.ops.gather(self.user_table, user_local, True)
        logq = self.logq[b["item"]] if self.logq is not None else None
        loss, du, dv = self.ops.loss_and_grads(self.kind, u, v, b["target"], b["item"], b["pos"], logq,
                                               self.num_negatives, 1.0, 1.0)
        if next_b is not None:
            self.prefetch(next_b)                 # the GPU is busy with the sweeps just queued
        dv_owned = ex.push(dv)
        jobs = [lambda: self.ops.update(self.optimizer, self.item_table, self.state["item"], ex.local_ids, dv_owned, True,
                                        self.steps, self.lr),
                lambda: self.ops.update(self.optimizer, self.user_table, self.state["user"], user_local, du, True,
                                        self.steps, self.lr)]
        if self.item_table.is_cuda:
            from .optim import run_table_jobs

            run_table_jobs(jobs)                      # the two tables update side by side (two streams)
        else:
            for job in jobs:
                job()
        return loss


class ShardedIndex:
    """Exact top-k over a row-sharded catalog; bit-identical to the single-GPU result.  Local row l of
    this rank's shard is global item row ``l * stride + offset`` (round-robin rows: stride = world,
    offset = rank; contiguous blocks: stride = 1, offset = first row)."""

    def __init__(self, item_shard: torch.Tensor, offset: int, num_items: int, *, stride: int = 1, ops=None, mf=None) -> None:
        self.items, self.offset, self.stride, self.num_items = item_shard, int(offset), int(stride), num_items
        if ops is None:
            import importlib

            ops = HipOps(mf if mf is not None else importlib.import_module(__package__))
        self.ops = ops

    def _localise(self, ids: torch.Tensor) -> torch.Tensor:
        """Global item rows -> rows of this shard (-1: not ours, ignored by the scan)."""
        d = ids - self.offset
        ours = (d >= 0) & (torch.remainder(d, self.stride) == 0)
        return torch.where(ours, torch.div(d, self.stride, rounding_mode="floor"), torch.full_like(d, -1))

    def search(self, queries: torch.Tensor, top_k: int, *, exclude_csr=None):
        world, rank = dist.get_world_size(), dist.get_rank()  # noqa: F841
        q, d = queries.shape
        all_q = torch.empty(world * q, d, dtype=queries.dtype, device=queries.device)
        dist.all_gather_into_tensor(all_q, queries.contiguous())
        csr = None
        key = None if exclude_csr is None else (exclude_csr[0].data_ptr(), exclude_csr[1].data_ptr(), exclude_csr[1].numel(), q)
        if key is not None and getattr(self, "_csr_key", None) == key:
            csr = self._csr_all                       # same exclusion lists as the last call: gathered once
        elif exclude_csr is not None:
            off, ids = exclude_csr
            n_loc = torch.tensor([ids.numel()], dtype=torch.int64, device=queries.device)
            n_all = [torch.empty_like(n_loc) for _ in range(world)]
            dist.all_gather(n_all, n_loc)
            n_all = [int(x) for x in n_all]                          # host sync: list lengths
            cap = max(max(n_all), 1)
            padded = torch.zeros(cap, dtype=torch.int64, device=queries.device)
            padded[: ids.numel()] = ids
            all_ids = torch.empty(world * cap, dtype=torch.int64, device=queries.device)
            dist.all_gather_into_tensor(all_ids, padded)
            all_off = torch.empty(world * (q + 1), dtype=torch.int64, device=queries.device)
            dist.all_gather_into_tensor(all_off, off.contiguous())
            pieces, offs, base = [], [torch.zeros(1, dtype=torch.int64, device=queries.device)], 0
            for r in range(world):
                pieces.append(all_ids[r * cap: r * cap + n_all[r]])
                offs.append(all_off[r * (q + 1) + 1: (r + 1) * (q + 1)] + base)
                base += n_all[r]
            ids_all = torch.cat(pieces) if base else torch.zeros(1, dtype=torch.int64, device=queries.device)
            csr = (torch.cat(offs), self._localise(ids_all) if base else ids_all - 1)
            self._csr_key, self._csr_all, self._csr_src = key, csr, exclude_csr   # (keeps the source tensors alive)
        ps, pi = self.ops.topk(all_q, self.items, top_k, csr, 0)                 # [world * q, k], local rows
        pi = torch.where(pi >= 0, pi * self.stride + self.offset, pi)            # -> global rows, before the merge
        rs = torch.empty_like(ps)
        ri = torch.empty_like(pi)
        dist.all_to_all_single(rs, ps.contiguous())                              # block g -> rank g
        dist.all_to_all_single(ri, pi.contiguous())
        return self.ops.merge(rs.reshape(world, q, top_k), ri.reshape(world, q, top_k), top_k)
