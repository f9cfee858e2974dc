"""Multi-GPU hot path: one process per GPU; exchanges by RCCL over xGMI on the compute stream (``RcclComm``,
``mf_comm_*``), or by ``torch.distributed`` (``TorchComm``: ``gloo`` in the CPU tests).

The reference has no explicit collective anywhere (implicit DDP of a
dense BERT); the sharding below is the north-star's and our design:

* **item table row-sharded, rows dealt round-robin** (rank r owns rows ``r, r + G, r + 2G, ...``):
  item popularity is heavy-tailed and id order often follows it, so contiguous blocks would send most
  of every batch to one owner (its gather, sort and update then set the step time of the whole job);
  **user table sharded in contiguous blocks** and the training pairs partitioned by user shard, so
  user rows never travel;
* **hash / bloom towers** (BASELINE config 5: 10 M users x 100 M items do not get a row each): both
  BUCKET tables are dealt round-robin and every id's ``num_hashes`` bucket rows go through the same
  exchange as item rows (an id's buckets live on arbitrary ranks, users included);
* every shard is **initialised on its own device** from a counter-based generator (``mf_init_rows``: a
  pure function of (seed, global row, column)) -- no rank ever holds a whole table (102 GB for
  100 M x 256) and the values do not depend on the number of ranks;
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

Local compute goes through an ``ops`` object (default :class:`HipOps`) and the exchanges through a ``comm``
object, so that the routing logic can be exercised on CPU with ``gloo`` by injecting the oracle.
No scaling curve has been measured: only one-GPU boxes were available.
"""
from __future__ import annotations

import ctypes
import math

import torch
import torch.distributed as dist

from . import _lib


# --------------------------------------------------------------------------- exchanges ---
class TorchComm:
    """Exchanges through ``torch.distributed`` (the CPU tests' ``gloo``; also works with ``nccl``, at two
    cross-stream joins per collective)."""

    def __init__(self) -> None:
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def rows(self, x: torch.Tensor, send_counts: list[int], recv_counts: list[int]) -> torch.Tensor:
        out = torch.empty((sum(recv_counts),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_to_all_single(out, x.contiguous(), output_split_sizes=recv_counts, input_split_sizes=send_counts)
        return out

    def counts(self, send: torch.Tensor) -> torch.Tensor:
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send)
        return recv

    def gather(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        return out

    def equal(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(x)
        dist.all_to_all_single(out, x.contiguous())
        return out


class RcclComm:
    """Exchanges by RCCL called from ``libmf_hip.so`` on the CURRENT stream (``mf_comm_*``): no second stream, no
    event joins.  The communicator is bootstrapped through the already initialised ``torch.distributed`` group
    (rank 0's 128-byte id is broadcast as a Python object)."""

    def __init__(self, device) -> None:
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        lib = _lib.lib()
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(lib.mf_comm_unique_id(buf))
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=0, device=torch.device(device))
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.mf_comm_create(self.world, self.rank, ctypes.create_string_buffer(box[0], 128), ctypes.byref(handle)))
        self.handle = handle

    def __del__(self) -> None:
        try:
            if getattr(self, "handle", None):
                _lib.lib().mf_comm_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def rows(self, x: torch.Tensor, send_counts: list[int], recv_counts: list[int]) -> torch.Tensor:
        x = x.contiguous()
        out = torch.empty((sum(recv_counts),) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        row_bytes = x.element_size() * math.prod(x.shape[1:])        # (1-D: one element per row)
        arr = ctypes.c_int64 * self.world
        _lib.check(_lib.lib().mf_comm_all_to_all_rows(self.handle, _lib.ptr(x) if x.numel() else None, arr(*send_counts),
                                                      _lib.ptr(out) if out.numel() else None, arr(*recv_counts), row_bytes,
                                                      _lib.stream_ptr()))
        return out

    def counts(self, send: torch.Tensor) -> torch.Tensor:
        return self.rows(send.reshape(self.world, 1), [1] * self.world, [1] * self.world).reshape(self.world)

    def gather(self, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        out = torch.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().mf_comm_all_gather(self.handle, _lib.ptr(x), _lib.ptr(out), x.numel() * x.element_size(),
                                                 _lib.stream_ptr()))
        return out

    def equal(self, x: torch.Tensor) -> torch.Tensor:
        per = x.shape[0] // self.world
        return self.rows(x, [per] * self.world, [per] * self.world)


def default_comm(device):
    """RCCL on the compute stream for GPU tensors (``RcclComm``), ``torch.distributed`` otherwise.

    The C-side communicator has only ever run on ONE rank where this was built (one GPU per box), so at world > 1 it is
    checked before it is trusted: every rank exchanges a small all-to-all and an all-gather through it and compares with
    ``torch.distributed``'s answer; unless ALL ranks agree the job uses ``TorchComm`` -- the same RCCL collectives issued by
    torch, a few cross-stream joins slower, never a CPU path.  ``MF_COMM=rccl`` / ``torch`` forces one or the other."""
    import os
    import warnings

    if torch.device(device).type != "cuda":
        return TorchComm()
    mode = os.environ.get("MF_COMM", "auto")
    if mode == "torch":
        return TorchComm()
    if mode == "rccl" or dist.get_world_size() == 1:
        return RcclComm(device)
    ok, comm, why = 1, None, ""
    try:
        comm = RcclComm(device)
        w, r = comm.world, comm.rank
        ref = TorchComm()
        send = torch.arange(w, dtype=torch.int64, device=device) * 1000 + r
        counts = [(r + j) % 3 + 1 for j in range(w)]                        # rows this rank sends to rank j
        recv_counts = [(j + r) % 3 + 1 for j in range(w)]                   # rows rank j sends here
        rows = torch.arange(sum(counts) * 4, dtype=torch.float32, device=device).reshape(-1, 4) + 100.0 * r
        same = (torch.equal(comm.counts(send), ref.counts(send)) and torch.equal(comm.gather(send), ref.gather(send))
                and torch.equal(comm.rows(rows, counts, recv_counts), ref.rows(rows, counts, recv_counts)))
        if not same:
            ok, why = 0, "self-test mismatch"
    except Exception as e:  # noqa: BLE001  (whatever the C side or RCCL raised: fall back together)
        ok, why = 0, repr(e)
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        return comm
    warnings.warn(f"RcclComm not used ({why or 'another rank failed its self-test'}); falling back to torch.distributed collectives",
                  stacklevel=2)
    return TorchComm()


# ----------------------------------------------------------------------- local compute ---
class HipOps:
    """Local compute on the GPU through libmf_hip.so (the product path)."""

    def __init__(self, mf) -> None:
        self.mf = mf

    def init_rows(self, n_local, d, row_start, row_stride, seed, std, device):
        out = torch.empty(n_local, d, dtype=torch.float32, device=device)
        _lib.check(_lib.lib().mf_init_rows(out.data_ptr(), n_local, d, row_start, row_stride, seed, std, _lib.stream_ptr()))
        return out

    def gather(self, table, ids, normalize, want_inv=False):
        out = torch.empty(ids.numel(), table.shape[1], dtype=torch.float32, device=table.device)
        inv = torch.empty(ids.numel(), dtype=torch.float32, device=table.device) if want_inv else None
        _lib.check(_lib.lib().mf_gather_rows(table.data_ptr(), table.shape[0], table.shape[1], ids.data_ptr(), ids.numel(),
                                             int(normalize), out.data_ptr(), _lib.ptr(inv), _lib.stream_ptr()))
        return (out, inv) if want_inv else out

    def stable_argsort(self, keys):
        """(perm int64, sorted keys) of a stable sort of small non-negative int64 keys (``mf_sort_keys``)."""
        lib = _lib.lib()
        n = keys.numel()
        perm = torch.empty(n, dtype=torch.int32, device=keys.device)
        sk = torch.empty(n, dtype=torch.int64, device=keys.device)
        ws = _lib.workspace(lib.mf_sort_ws_bytes(n), keys.device)
        _lib.check(lib.mf_sort_keys(keys.data_ptr(), n, perm.data_ptr(), sk.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        return perm.long(), sk

    def hash_buckets(self, ids, num_hashes, seed, num_buckets):
        out = torch.empty(ids.numel() * num_hashes, dtype=torch.int64, device=ids.device)
        _lib.check(_lib.lib().mf_hash_buckets(ids.data_ptr(), ids.numel(), num_hashes, seed, num_buckets, out.data_ptr(),
                                              _lib.stream_ptr()))
        return out

    def bloom_forward(self, bucket_rows, num_hashes):
        """[n * H, d] fetched bucket rows -> unit rows [n, d] and 1 / norm [n] (sum in hash order, then the tower's
        L2-normalisation through the gather kernel with an identity index)."""
        n = bucket_rows.shape[0] // num_hashes
        summed = bucket_rows.view(n, num_hashes, -1).sum(dim=1) if num_hashes > 1 else bucket_rows
        return self.gather(summed.contiguous(), torch.arange(n, device=bucket_rows.device), True, want_inv=True)

    def bloom_backward(self, unit, inv, grad, num_hashes):
        graw = torch.empty_like(grad)
        _lib.check(_lib.lib().mf_normalize_backward(unit.data_ptr(), inv.data_ptr(), grad.contiguous().data_ptr(), unit.shape[0],
                                                    unit.shape[1], graw.data_ptr(), _lib.stream_ptr()))
        return graw.repeat_interleave(num_hashes, dim=0) if num_hashes > 1 else graw

    def loss_and_grads(self, kind, u, v, target, item_idx, pos_idx, logq_table, num_negatives, sigma, margin):
        """``logq_table``: logQ of every GLOBAL item row (looked up by ``item_idx`` inside the kernel), or None."""
        u = u.detach().requires_grad_()
        v = v.detach().requires_grad_()
        fn = getattr(self.mf.losses, kind)(num_negatives=num_negatives, sigma=sigma, margin=margin)
        loss = fn(u, v, target, item_idx=item_idx, pos_idx=pos_idx, logq_table=logq_table)
        one = getattr(self, "_one", None)
        if one is None or one.device != u.device:
            one = self._one = torch.ones((), device=u.device)
        loss.backward(one)
        return loss.detach(), u.grad, v.grad

    def update(self, optimizer, table, state, ids, grad, normalized, step, hyper):
        lib = _lib.lib()
        n, d = ids.numel(), table.shape[1]
        if n == 0:
            return
        ws = _lib.workspace(lib.mf_update_ws_bytes(n, d), table.device)
        if optimizer == "sgd":
            _lib.check(lib.mf_update_sgd(table.data_ptr(), table.shape[0], d, ids.data_ptr(), n, grad.data_ptr(),
                                         int(normalized), hyper["lr"], hyper["weight_decay"], ws.data_ptr(), ws.numel(),
                                         _lib.stream_ptr()))
        else:
            b1, b2 = hyper["betas"]
            _lib.check(lib.mf_update_adam(table.data_ptr(), state["m"].data_ptr(), state["v"].data_ptr(), table.shape[0], d,
                                          ids.data_ptr(), n, grad.data_ptr(), int(normalized), step, None, hyper["lr"], b1, b2,
                                          hyper["eps"], hyper["weight_decay"], ws.data_ptr(), ws.numel(), _lib.stream_ptr()))

    def topk(self, queries, items, k, exclude_csr, idx_base):
        # one ItemIndex per shard tensor: its derived copies (bf16 rows, blocked rows) and workspaces are built once.
        # Like the reference's index (a table written by get_index), it is a SNAPSHOT: drop_indexes() after the rows change.
        key = (items.data_ptr(), tuple(items.shape), int(idx_base))
        cache = self.__dict__.setdefault("_indexes", {})
        index = cache.get(key)
        if index is None:
            if len(cache) >= 4:
                cache.clear()
            index = cache[key] = self.mf.retrieval.ItemIndex(items, idx_base=idx_base)
        return index.search(queries, k, exclude_csr=exclude_csr)

    def drop_indexes(self) -> None:
        self.__dict__.pop("_indexes", None)

    def merge(self, part_scores, part_rows, k):
        return self.mf.retrieval.merge_topk(part_scores, part_rows, k)

    def pack(self, scores, rows, stride, offset):
        """(scores, LOCAL rows) -> one int64 per entry with the GLOBAL row (``mf_topk_pack``): one exchange, not two."""
        out = torch.empty(rows.shape, dtype=torch.int64, device=rows.device)
        _lib.check(_lib.lib().mf_topk_pack(scores.data_ptr(), rows.data_ptr(), rows.numel(), int(stride), int(offset), out.data_ptr(),
                                           _lib.stream_ptr()))
        return out

    def merge_packed(self, packed, world, q, k):
        scores = torch.empty(q, k, dtype=torch.float32, device=packed.device)
        rows = torch.empty(q, k, dtype=torch.int64, device=packed.device)
        _lib.check(_lib.lib().mf_topk_merge_packed(packed.data_ptr(), world, q, k, scores.data_ptr(), rows.data_ptr(), _lib.stream_ptr()))
        return scores, rows


def optimizer_hyper(optimizer: str, lr: float | None = None, **over) -> dict:
    """The hyper-parameters of ``optim.SparseSGD`` / ``optim.RowAdam`` (their constructor defaults), in one place."""
    if optimizer == "sgd":
        h = {"lr": 1e-2, "weight_decay": 0.0}
    else:
        h = {"lr": 1e-4, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0.01}
    if lr is not None:
        h["lr"] = lr
    h.update(over)
    return h


def shard_bounds(n_rows: int, world: int, rank: int) -> tuple[int, int]:
    per = (n_rows + world - 1) // world
    return min(rank * per, n_rows), min((rank + 1) * per, n_rows)


def cyclic_rows(n_rows: int, world: int, rank: int) -> int:
    """Rows of a table dealt round-robin that land on ``rank``."""
    return (n_rows - rank + world - 1) // world if n_rows > rank else 0


class RowExchange:
    """Routes a list of global row ids to their owning ranks and back (rows dealt round-robin:
    owner = id mod world, local row = id div world)."""

    def __init__(self, ids: torch.Tensor, comm, ops=None) -> None:
        self.comm, world = comm, comm.world
        self.ids = ids                                               # keeps the routed ids (and their storage) alive
        owner = torch.remainder(ids, world)
        # batch position of every sent slot: a stable sort by owner (on the GPU the library's rank sort -- three small
        # launches -- instead of torch's radix sort + bincount: ~250 us of kernels on the plan stream at n = 16,384)
        if ops is not None and hasattr(ops, "stable_argsort") and ids.is_cuda:
            self.order, sorted_owner = ops.stable_argsort(owner)
            bounds = torch.searchsorted(sorted_owner, torch.arange(world + 1, device=ids.device, dtype=sorted_owner.dtype))
            send = bounds[1:] - bounds[:-1]
        else:
            self.order = torch.argsort(owner, stable=True)
            send = torch.bincount(owner, minlength=world)
        recv = comm.counts(send)
        # RCCL needs the split sizes on the host: the one host sync of the step
        self.send_counts, self.recv_counts = send.tolist(), recv.tolist()
        # rows of MY shard that the others (and I) asked for, grouped by requesting rank
        self.local_ids = torch.div(comm.rows(ids[self.order], self.send_counts, self.recv_counts), world, rounding_mode="floor")

    def fetch(self, rows_for_requests: torch.Tensor) -> torch.Tensor:
        """owner -> requester: rows gathered for ``local_ids`` come back in batch order."""
        got = self.comm.rows(rows_for_requests, self.recv_counts, self.send_counts)
        out = torch.empty_like(got)
        out[self.order] = got
        return out

    def push(self, per_batch_rows: torch.Tensor) -> torch.Tensor:
        """requester -> owner: one row per batch slot, delivered aligned with ``local_ids``."""
        return self.comm.rows(per_batch_rows[self.order], self.send_counts, self.recv_counts)


class _Plan:
    """The exchange plans of one batch (+ the event that orders a side-stream build before its use)."""

    def __init__(self, key, item: RowExchange, user: RowExchange | None, ready, built_at: int, buckets) -> None:
        self.key, self.item, self.user, self.ready, self.built_at, self.buckets = key, item, user, ready, built_at, buckets


class ShardedTrainer:
    """Row-sharded tables + the training step described in the module docstring.  ``num_hashes > 0``: hash / bloom
    towers -- ``num_users`` / ``num_items`` are then BUCKET counts and ids may be arbitrary int64."""

    def __init__(self, mf, device, optimizer: str, num_negatives: int, *, num_users: int, num_items: int, dim: int,
                 logq: torch.Tensor | None = None, kind: str = "InfomationNoiseContrastiveEstimationLoss",
                 lr: float | None = None, ops=None, comm=None, seed: int = 0, num_hashes: int = 0, hash_seed: int = 0) -> None:
        self.ops = ops if ops is not None else HipOps(mf)
        self.comm = comm if comm is not None else default_comm(device)
        self.rank, self.world = self.comm.rank, self.comm.world
        self.optimizer, self.num_negatives, self.kind = optimizer, num_negatives, kind
        self.hyper = optimizer_hyper(optimizer, lr)
        self.num_users, self.num_items, self.dim = num_users, num_items, dim
        self.num_hashes, self.hash_seed = int(num_hashes), int(hash_seed)
        std = 1.0 / (dim * max(self.num_hashes, 1)) ** 0.5
        # every shard is generated in place (seed + 1: the item table's stream)
        if self.num_hashes:                                       # both bucket tables dealt round-robin
            self.user_lo, self.user_hi = 0, num_users
            self.user_table = self.ops.init_rows(cyclic_rows(num_users, self.world, self.rank), dim, self.rank, self.world, seed,
                                                 std, device)
        else:                                                     # contiguous user blocks: user rows never travel
            self.user_lo, self.user_hi = shard_bounds(num_users, self.world, self.rank)
            self.user_table = self.ops.init_rows(self.user_hi - self.user_lo, dim, self.user_lo, 1, seed, std, device)
        self.item_table = self.ops.init_rows(cyclic_rows(num_items, self.world, self.rank), dim, self.rank, self.world, seed + 1,
                                             std, device)           # rows rank, rank + world, ...
        self.state = {name: {"m": torch.zeros_like(t), "v": torch.zeros_like(t)}
                      for name, t in (("user", self.user_table), ("item", self.item_table))}
        self.logq = logq
        self.steps = 0
        self._plans: dict = {}
        self._plan_stream = None

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
    @staticmethod
    def _key(b) -> tuple:
        return (b["item"].data_ptr(), b["item"]._version, b["item"].numel(), b["user"].data_ptr(), b["user"]._version)

    def _build(self, b) -> tuple:
        if self.num_hashes:
            ib = self.ops.hash_buckets(b["item"], self.num_hashes, self.hash_seed + 1, self.num_items)
            ub = self.ops.hash_buckets(b["user"], self.num_hashes, self.hash_seed, self.num_users)
            return RowExchange(ib, self.comm, self.ops), RowExchange(ub, self.comm, self.ops), (ub, ib)
        return RowExchange(b["item"], self.comm, self.ops), None, None

    def _plan(self, b) -> _Plan:
        """The routing plan of a batch.  Building one costs a host sync (the split sizes); ``prefetch`` moves that sync
        off the critical path.  A prefetched plan is only used for the very tensors it was built from (same storage, same
        version: the plan holds a reference, so the allocator cannot hand that address to another batch meanwhile)."""
        key = self._key(b)
        hit = self._plans.pop(key, None)
        # whatever else was prefetched and not consumed by now (epoch end, skipped batch) is stale: drop it
        for k in [k for k, p in self._plans.items() if p.built_at < self.steps - 1]:
            del self._plans[k]
        if hit is None:
            item, user, buckets = self._build(b)
            return _Plan(key, item, user, None, self.steps, buckets)
        if hit.ready is not None:                  # built on the side stream: order it before our use
            cur = torch.cuda.current_stream()
            cur.wait_event(hit.ready)
            for ex in (hit.item, hit.user):
                if ex is not None:
                    for t in (ex.order, ex.local_ids):
                        t.record_stream(cur)
        return hit

    def prefetch(self, next_b, after: "torch.cuda.Event | None" = None) -> None:
        """Build the plan of the NEXT batch now, on a side stream: its small kernels, its tiny exchanges and the host
        read of the split sizes run beside the current step's MFMA sweeps instead of stalling the start of the next
        step (ids of the next batch are known: prefetching loader).  The ids may have just been produced on the current
        stream (device sampler): the side stream first waits for ``after`` -- an event recorded once they were queued
        (``step`` records one on entry: its ``next_b`` argument exists by then, its own sweeps do not) -- or, without
        one, for everything queued on the current stream so far.  Every rank must call this at the same point of its
        step (collective order)."""
        key = self._key(next_b)
        if key in self._plans:
            return
        ids = next_b["item"]
        if not ids.is_cuda:
            item, user, buckets = self._build(next_b)
            self._plans[key] = _Plan(key, item, user, None, self.steps, buckets)
            return
        cur = torch.cuda.current_stream()
        if self._plan_stream is None:
            # high priority: its own hardware queue, so the small plan kernels are dispatched beside the sweeps
            self._plan_stream = torch.cuda.Stream(device=ids.device, priority=-1)
        if after is not None:
            self._plan_stream.wait_event(after)     # the ids were queued before this event
        else:
            self._plan_stream.wait_stream(cur)      # conservative: everything queued so far
        with torch.cuda.stream(self._plan_stream):
            item, user, buckets = self._build(next_b)
            ready = torch.cuda.Event()
            ready.record(self._plan_stream)
        for t in (next_b["item"], next_b["user"]):
            t.record_stream(self._plan_stream)
        self._plans[key] = _Plan(key, item, user, ready, self.steps, buckets)

    # -- one step -----------------------------------------------------------------------------
    def step(self, b, next_b=None) -> torch.Tensor:
        """``b``: ``user`` (global rows inside this rank's user shard; any ids with hashed towers), ``item`` (2B
        global rows: positives then negatives), ``target``, ``pos``.  ``next_b``: the batch after it, if known
        (its exchange plan is then prefetched behind this step's compute)."""
        self.steps += 1
        entry = None
        if next_b is not None and b["item"].is_cuda:
            entry = torch.cuda.Event()
            entry.record()                        # next_b's ids are queued by now; this step's sweeps are not
        plan = self._plan(b)
        ops, H = self.ops, self.num_hashes
        if H:
            v, v_inv = ops.bloom_forward(plan.item.fetch(ops.gather(self.item_table, plan.item.local_ids, False)), H)
            u, u_inv = ops.bloom_forward(plan.user.fetch(ops.gather(self.user_table, plan.user.local_ids, False)), H)
        else:
            user_local = b["user"] - self.user_lo
            v = plan.item.fetch(ops.gather(self.item_table, plan.item.local_ids, True))
            u = ops.gather(self.user_table, user_local, True)
        loss, du, dv = ops.loss_and_grads(self.kind, u, v, b["target"], b["item"], b["pos"], self.logq, self.num_negatives, 1.0, 1.0)
        if next_b is not None:
            self.prefetch(next_b, after=entry)    # the GPU is busy with the sweeps just queued
        if H:
            dv_owned = plan.item.push(ops.bloom_backward(v, v_inv, dv, H))
            du_owned = plan.user.push(ops.bloom_backward(u, u_inv, du, H))
            ops.update(self.optimizer, self.item_table, self.state["item"], plan.item.local_ids, dv_owned, False, self.steps, self.hyper)
            ops.update(self.optimizer, self.user_table, self.state["user"], plan.user.local_ids, du_owned, False, self.steps, self.hyper)
        else:
            dv_owned = plan.item.push(dv)
            ops.update(self.optimizer, self.item_table, self.state["item"], plan.item.local_ids, dv_owned, True, self.steps, self.hyper)
            ops.update(self.optimizer, self.user_table, self.state["user"], user_local, du, True, self.steps, self.hyper)
        return loss


class ShardedIndex:
    """Exact top-k over a row-sharded catalog; bit-identical to the single-GPU result.  Local row l of
    this rank's shard is global item row ``l * stride + offset`` (round-robin rows: stride = world,
    offset = rank; contiguous blocks: stride = 1, offset = first row)."""

    def __init__(self, item_shard: torch.Tensor, offset: int, num_items: int, *, stride: int = 1, ops=None, mf=None,
                 comm=None) -> None:
        self.items, self.offset, self.stride, self.num_items = item_shard, int(offset), int(stride), num_items
        if ops is None:
            import importlib

            ops = HipOps(mf if mf is not None else importlib.import_module(__package__))
        self.ops = ops
        self.comm = comm if comm is not None else default_comm(item_shard.device)

    def refresh(self) -> None:
        """The shard's rows changed (training went on): rebuild the derived search copies at the next search."""
        if hasattr(self.ops, "drop_indexes"):
            self.ops.drop_indexes()

    def _localise(self, ids: torch.Tensor) -> torch.Tensor:
        """Global item rows -> rows of this shard (-1: not ours, ignored by the scan)."""
        d = ids - self.offset
        ours = (d >= 0) & (torch.remainder(d, self.stride) == 0)
        return torch.where(ours, torch.div(d, self.stride, rounding_mode="floor"), torch.full_like(d, -1))

    def search(self, queries: torch.Tensor, top_k: int, *, exclude_csr=None):
        comm = self.comm
        world = comm.world
        q, d = queries.shape
        all_q = comm.gather(queries)
        csr = None
        key = None if exclude_csr is None else (exclude_csr[0].data_ptr(), exclude_csr[1].data_ptr(), exclude_csr[1].numel(), q)
        if key is not None and getattr(self, "_csr_key", None) == key:
            csr = self._csr_all                       # same exclusion lists as the last call: gathered once
        elif exclude_csr is not None:
            off, ids = exclude_csr
            n_loc = torch.tensor([ids.numel()], dtype=torch.int64, device=queries.device)
            n_all = [int(x) for x in comm.gather(n_loc).tolist()]       # host sync: list lengths
            cap = max(max(n_all), 1)
            padded = torch.zeros(cap, dtype=torch.int64, device=queries.device)
            padded[: ids.numel()] = ids
            all_ids = comm.gather(padded)
            all_off = comm.gather(off.contiguous())
            pieces, offs, base = [], [torch.zeros(1, dtype=torch.int64, device=queries.device)], 0
            for r in range(world):
                pieces.append(all_ids[r * cap: r * cap + n_all[r]])
                offs.append(all_off[r * (q + 1) + 1: (r + 1) * (q + 1)] + base)
                base += n_all[r]
            ids_all = torch.cat(pieces) if base else torch.zeros(1, dtype=torch.int64, device=queries.device)
            csr = (torch.cat(offs), self._localise(ids_all) if base else ids_all - 1)
            self._csr_key, self._csr_all, self._csr_src = key, csr, exclude_csr   # (keeps the source tensors alive)
        ps, pi = self.ops.topk(all_q, self.items, top_k, csr, 0)                 # [world * q, k], local rows
        if hasattr(self.ops, "pack"):
            # scores and global rows as ONE int64 per entry: one launch for the row mapping, one exchange, merged as they are
            packed = comm.equal(self.ops.pack(ps, pi, self.stride, self.offset))
            return self.ops.merge_packed(packed, world, q, top_k)
        pi = torch.where(pi >= 0, pi * self.stride + self.offset, pi)            # -> global rows, before the merge
        rs = comm.equal(ps)                                                      # block g -> rank g
        ri = comm.equal(pi)
        return self.ops.merge(rs.reshape(world, q, top_k), ri.reshape(world, q, top_k), top_k)
