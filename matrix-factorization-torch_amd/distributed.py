"""Multi-GPU hot path: one process per GPU; exchanges by RCCL over xGMI on the compute stream (``RcclComm``,
``mf_comm_*``), or by ``torch.distributed`` (``TorchComm``: ``gloo`` in the CPU tests).

The reference has no explicit collective anywhere (implicit DDP of a dense BERT, examples dealt to the
ranks by ``sharding_filter``, xfmr_rec/data/lightning.py:109; one worker by default, xfmr_rec/ray.py:40); the sharding
below is the north-star's and our design:

* **both tables row-sharded, rows dealt round-robin** (rank r owns rows ``r, r + G, r + 2G, ...``): item popularity is
  heavy-tailed and id order often follows it, so contiguous blocks would send most of every batch to one owner (its
  gather, sort and update then set the step time of the whole job).  A rank's batch may hold ARBITRARY users and items
  (example-sharded data, like the reference's loader): every id is routed to its owner and back (``RowExchange``).
  ``user_mode="partitioned"`` keeps the round-2 fast path -- the user table in contiguous blocks and the training pairs
  partitioned by user shard, so user rows never travel -- and a batch that breaks that promise RAISES (it used to gather
  a silent zero row); ``user_mode="replicated"`` is the split the north-star words: item corpus sharded, user table
  replicated, user-row gradients combined by a sparse all-gather of (row, gradient) and applied by every rank;
* **hash / bloom towers** (BASELINE config 5: 10 M users x 100 M items do not get a row each): both
  BUCKET tables are dealt round-robin and every id's ``num_hashes`` bucket rows go through the same
  exchange (an id's buckets live on arbitrary ranks);
* every shard is **initialised on its own device** from a counter-based generator (``mf_init_rows``: a
  pure function of (seed, global row, column)) -- no rank ever holds a whole table (102 GB for
  100 M x 256) and the values do not depend on the number of ranks;
* **training step**: each rank needs the rows of its own batch (B users, B positives + B sampled negatives).  Rows are
  fetched from their owners with one all-to-all of ids and one all-to-all of rows per table (8 MB per rank at B = 8192,
  d = 128 -- latency-bound on xGMI, so direct all-to-all, never a ring all-reduce of a dense table: the north-star's
  "all-reduce of user gradients" would move the whole 83 MB user table every step), the score / loss kernels run on the
  local B x 2B block (in-batch negatives stay local, exactly like the reference, which has no cross-rank gather),
  row gradients go back to their owners with one more all-to-all and each owner applies ONE sparse update per table and
  step (duplicates summed in (rank, batch) order: deterministic);
* **exchange plans**: RCCL wants the per-peer row counts on the host.  A plan built AHEAD (``prefetch``: the ids of the
  next batch are known, the host read happens beside the current step's sweeps) uses exact counts; a step whose plan
  was not prefetched uses **capacity-padded** exchanges -- equal splits of ``capacity`` rows per peer, padding ids -1,
  which the gather answers with zeros and the update skips -- and never reads anything back (no host sync);
* **retrieval**: queries are all-gathered, every rank scans ITS shard for all queries
  (``mf_topk`` on local rows, mapped to global rows before they leave), the partial top-k travel back to the query's rank by
  all-to-all and are merged exactly (``mf_topk_merge``): bit-identical to the
  single-GPU result.

Local compute goes through an ``ops`` object (default :class:`HipOps`) and the exchanges through a ``comm``
object, so that the routing logic can be exercised on CPU with ``gloo`` by injecting the oracle.
No scaling curve has been measured by the builder: only one-GPU boxes were available.
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

    transport = "torch"
    rccl_ranks = None

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

    def for_second_stream(self):
        """torch's process group orders its collectives itself (its own stream and events): one object serves every stream."""
        return self


class RcclComm:
    """Exchanges by RCCL called from ``libmf_hip.so`` on the CURRENT stream (``mf_comm_*``): no second stream, no
    event joins.  The communicator is bootstrapped through the already initialised ``torch.distributed`` group
    (rank 0's 128-byte id is broadcast as a Python object, together with rank 0's verdict on getting it: a rank
    that cannot create the id must not leave the others waiting in another collective)."""

    transport = "mf_comm"

    def __init__(self, device) -> None:
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        lib = _lib.lib()
        buf = ctypes.create_string_buffer(128)
        err = ""
        if self.rank == 0:
            rc = lib.mf_comm_unique_id(buf)
            if rc != 0:
                err = lib.mf_last_error().decode()
        box = [bytes(buf.raw), err]
        dist.broadcast_object_list(box, src=0, device=torch.device(device))
        if box[1]:
            raise _lib.MfHipError(f"rank 0 could not create an RCCL id: {box[1]}")
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.mf_comm_create(self.world, self.rank, ctypes.create_string_buffer(box[0], 128), ctypes.byref(handle)))
        self.handle = handle
        self.device = device

    def for_second_stream(self):
        """A SECOND communicator (its own ``ncclCommInitRank``, collective: every rank calls this at the same point).  One
        ``ncclComm`` must not be driven from two streams at once -- RCCL serialises a communicator's operations only within
        a stream -- and ``ShardedTrainer.prefetch`` builds the next plan on a side stream while the step's own exchanges run
        on the compute stream."""
        return RcclComm(self.device)

    @property
    def rccl_ranks(self) -> int:
        return int(_lib.lib().mf_comm_world(self.handle))

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


def _all_agree(ok: bool, device) -> bool:
    """True iff ``ok`` on EVERY rank (one tiny all-reduce through torch.distributed: every rank must call it at the same point)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def default_comm(device):
    """RCCL on the compute stream for GPU tensors (``RcclComm``), ``torch.distributed`` otherwise.

    The C-side communicator has only ever run on ONE rank where this was built (one GPU per box), so at world > 1 it is
    checked before it is trusted, and every stage is AGREED on by all ranks before the next collective is entered (a rank
    that raised while the others sat in a different collective would hang the job -- ADVICE r2):

    1. every rank reports whether ``libmf_hip.so`` found RCCL at all (``mf_comm_source``); unless all did, the whole job
       uses ``TorchComm`` -- the same RCCL collectives issued by torch, a few cross-stream joins slower, never a CPU path;
    2. rank 0's id travels together with rank 0's verdict on creating it; all ranks agree on the outcome of
       ``ncclCommInitRank``;
    3. a small all-to-all of counts, an all-gather and a ragged all-to-all of rows are each compared with
       ``torch.distributed``'s answer and agreed on one by one.
    A failure in 2 or 3 leaves RCCL in an unknown state: it is raised on EVERY rank (the launcher sees a non-zero exit)
    instead of being papered over.  ``MF_COMM=rccl`` / ``torch`` forces one or the other without the checks."""
    import os
    import warnings

    if torch.device(device).type != "cuda":
        return TorchComm()
    mode = os.environ.get("MF_COMM", "auto")
    if mode == "torch":
        return TorchComm()
    if mode == "rccl" or dist.get_world_size() == 1:
        return RcclComm(device)
    source = _lib.lib().mf_comm_source().decode()
    if not _all_agree(source != "not loaded", device):
        warnings.warn(f"RcclComm not used (librccl not loadable from libmf_hip.so on some rank; here: {source}); "
                      "using torch.distributed collectives on every rank", stacklevel=2)
        return TorchComm()
    comm, why = None, ""
    try:
        comm = RcclComm(device)                 # (its broadcast carries rank 0's verdict: all ranks raise, or none)
    except Exception as e:  # noqa: BLE001
        why = f"communicator: {e!r}"
    if not _all_agree(comm is not None, device):
        raise _lib.MfHipError(f"RCCL communicator could not be created on every rank ({why or 'another rank failed'})")
    w, r = comm.world, comm.rank
    ref = TorchComm()
    send = torch.arange(w, dtype=torch.int64, device=device) * 1000 + r
    counts = [(r + j) % 3 + 1 for j in range(w)]                        # rows this rank sends to rank j
    recv_counts = [(j + r) % 3 + 1 for j in range(w)]                   # rows rank j sends here
    rows = torch.arange(sum(counts) * 4, dtype=torch.float32, device=device).reshape(-1, 4) + 100.0 * r
    for name, mine, theirs in (("counts", lambda: comm.counts(send), lambda: ref.counts(send)),
                               ("gather", lambda: comm.gather(send), lambda: ref.gather(send)),
                               ("rows", lambda: comm.rows(rows, counts, recv_counts), lambda: ref.rows(rows, counts, recv_counts))):
        ok, why = True, ""
        try:
            got = mine()
        except Exception as e:  # noqa: BLE001
            ok, why, got = False, repr(e), None
        want = theirs()                          # every rank enters torch's collective, whatever happened above
        if ok and not torch.equal(got, want):
            ok, why = False, "mismatch against torch.distributed"
        if not _all_agree(ok, device):
            raise _lib.MfHipError(f"RcclComm self-test '{name}' failed ({why or 'on another rank'}); set MF_COMM=torch to run "
                                  "on torch.distributed's collectives")
    return comm


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

    GROUP_MAX_N, GROUP_MAX_KEYS = 32768, 64

    def group_by_key(self, keys, nkeys):
        """(perm int64, sorted keys, bounds[nkeys + 1]) of a stable grouping by a small key (``mf_group_keys``: one launch, one
        workgroup), or None when the list / key range is beyond its limits."""
        n = keys.numel()
        if n > self.GROUP_MAX_N or nkeys > self.GROUP_MAX_KEYS:
            return None
        perm = torch.empty(n, dtype=torch.int32, device=keys.device)
        sk = torch.empty(n, dtype=torch.int64, device=keys.device)
        bounds = torch.empty(nkeys + 1, dtype=torch.int64, device=keys.device)
        _lib.check(_lib.lib().mf_group_keys(keys.data_ptr(), n, nkeys, perm.data_ptr(), sk.data_ptr(), bounds.data_ptr(), _lib.stream_ptr()))
        return perm.long(), sk, bounds

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

    def loss_and_grads(self, kind, u, v, target, item_idx, pos_idx, logq_table, num_negatives, sigma, margin, pos_csr=None):
        """``logq_table``: logQ of every GLOBAL item row (looked up by ``item_idx`` inside the kernel), or None.
        ``pos_csr``: (user ids, pos_off, pos_items) instead of the padded ``pos_idx``."""
        u = u.detach().requires_grad_()
        v = v.detach().requires_grad_()
        fn = getattr(self.mf.losses, kind)(num_negatives=num_negatives, sigma=sigma, margin=margin)
        loss = fn(u, v, target, item_idx=item_idx, pos_idx=pos_idx, logq_table=logq_table, pos_csr=pos_csr)
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
        # ONE ItemIndex for the shard tensor last searched: its derived copies (bf16 rows, blocked rows) and workspaces are
        # built once.  The entry holds the source tensor itself (so its address cannot be handed to another tensor while
        # the entry lives) and its version counter: an in-place change of the rows rebuilds the index.  Like the
        # reference's index (a table written by get_index) it is a SNAPSHOT of the rows at build time.
        hit = self.__dict__.get("_index")
        if hit is None or hit[0] is not items or hit[1] != items._version or hit[2] != int(idx_base):
            hit = self._index = (items, items._version, int(idx_base), self.mf.retrieval.ItemIndex(items, idx_base=idx_base))
        return hit[3].search(queries, k, exclude_csr=exclude_csr)

    def drop_indexes(self) -> None:
        self.__dict__.pop("_index", None)

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
    """Routes a list of row requests to their owning ranks and back (rows dealt round-robin: owner = id mod world,
    local row = id div world).  ``ids`` are the global rows; ``payload`` (default: the ids) is what travels to the owner
    -- e.g. ``id << 1 | table`` when one exchange serves two tables.

    ``capacity=None``: exact per-peer counts (RCCL needs them on the HOST: one host sync, which ``ShardedTrainer.prefetch``
    moves off the critical path).  ``capacity=C``: every peer gets a fixed block of C slots (padding payload -1), so
    nothing is read back; ``overflow`` is a device flag that is set when some peer's share did not fit (never for
    C >= len(ids)) -- the exchange is then incomplete and the caller must treat the step as invalid."""

    def __init__(self, ids: torch.Tensor, comm, ops=None, capacity: int | None = None, payload: torch.Tensor | None = None) -> None:
        # ``comm`` serves the plan's own exchanges (counts, requests); ``fetch`` / ``push`` use ``self.comm``, which the trainer
        # re-points at the compute stream's communicator when a plan built on the side stream is consumed
        self.comm, world = comm, comm.world
        self.ids = ids                                               # keeps the routed ids (and their storage) alive
        n = ids.numel()
        payload = ids if payload is None else payload
        owner = torch.remainder(ids, world)
        # batch position of every sent slot: a stable sort by owner (on the GPU the library's rank sort -- three small
        # launches -- instead of torch's radix sort + bincount: ~250 us of kernels on the plan stream at n = 16,384)
        grouped = ops.group_by_key(owner, world) if ops is not None and hasattr(ops, "group_by_key") and ids.is_cuda else None
        if grouped is not None:
            # one counting-sort launch of one workgroup (the all-pairs rank sort below took 100 us of every CU at n = 24,576)
            self.order, sorted_owner, bounds = grouped
        elif ops is not None and hasattr(ops, "stable_argsort") and ids.is_cuda:
            self.order, sorted_owner = ops.stable_argsort(owner)
            bounds = torch.searchsorted(sorted_owner, torch.arange(world + 1, device=ids.device, dtype=sorted_owner.dtype))
        else:
            self.order = torch.argsort(owner, stable=True)
            sorted_owner = owner[self.order]
            bounds = torch.cat([torch.zeros(1, dtype=torch.int64, device=ids.device), torch.cumsum(torch.bincount(owner, minlength=world), 0)])
        send = bounds[1:] - bounds[:-1]
        self.capacity = None if capacity is None else int(capacity)
        if self.capacity is None:
            recv = comm.counts(send)
            # RCCL needs the split sizes on the host: the one host sync of an exact plan
            self.send_counts, self.recv_counts = send.tolist(), recv.tolist()
            # requests for rows of MY shard, grouped by requesting rank
            self.requests = comm.rows(payload[self.order], self.send_counts, self.recv_counts)
            self.overflow = None
        else:
            cap = self.capacity
            pos = torch.arange(n, device=ids.device) - bounds[:-1][sorted_owner]          # position inside the owner's block
            last = world * cap                                                            # one spare slot swallows what does not fit
            self.slot = torch.where(pos < cap, sorted_owner * cap + pos, torch.full_like(pos, last))
            self.overflow = (send > cap).any()
            buf = torch.full((last + 1,), -1, dtype=payload.dtype, device=ids.device)
            buf[self.slot] = payload[self.order]
            self.requests = comm.equal(buf[:last])                                        # [world * cap], -1 = padding
        self.local_ids = torch.where(self.requests >= 0, torch.div(self.requests, world, rounding_mode="floor"),
                                     torch.full_like(self.requests, -1))

    def fetch(self, rows_for_requests: torch.Tensor) -> torch.Tensor:
        """owner -> requester: rows gathered for ``requests`` come back in batch order."""
        out = torch.empty((self.ids.numel(),) + tuple(rows_for_requests.shape[1:]), dtype=rows_for_requests.dtype,
                          device=rows_for_requests.device)
        if self.capacity is None:
            out[self.order] = self.comm.rows(rows_for_requests, self.recv_counts, self.send_counts)
        else:
            got = self.comm.equal(rows_for_requests)
            out[self.order] = got[self.slot.clamp_max(got.shape[0] - 1)]                  # (overflowed slots: flagged, garbage)
        return out

    def push(self, per_batch_rows: torch.Tensor) -> torch.Tensor:
        """requester -> owner: one row per batch slot, delivered aligned with ``requests`` (padding slots carry
        unspecified values: their request is -1 and every consumer skips them)."""
        if self.capacity is None:
            return self.comm.rows(per_batch_rows[self.order], self.send_counts, self.recv_counts)
        last = self.comm.world * self.capacity
        buf = torch.zeros((last + 1,) + tuple(per_batch_rows.shape[1:]), dtype=per_batch_rows.dtype, device=per_batch_rows.device)
        buf[self.slot] = per_batch_rows[self.order]
        return self.comm.equal(buf[:last])


class _Plan:
    """The exchange plans of one batch (+ the event that orders a side-stream build before its use)."""

    def __init__(self, key, item: RowExchange, user: RowExchange | None, ready, built_at: int, buckets, bad_users=None) -> None:
        self.key, self.item, self.user, self.ready, self.built_at, self.buckets = key, item, user, ready, built_at, buckets
        self.bad_users = bad_users          # device flag: ids outside this rank's user shard (partitioned mode)


class _LazyCheck:
    """A device-side error flag that is read WITHOUT stalling the step that produced it: its value is copied to pinned
    host memory behind the work that computes it, and looked at once that copy has completed (a later step, or
    ``ShardedTrainer.finish``)."""

    def __init__(self, step: int, message: str, flag: torch.Tensor) -> None:
        self.step, self.message = step, message
        if flag.is_cuda:
            self.host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            self.host.copy_(flag.reshape(1).to(torch.int32), non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host, self.event = flag.reshape(1).to(torch.int32), None

    def done(self) -> bool:
        return self.event is None or self.event.query()

    def wait(self) -> None:
        if self.event is not None:
            self.event.synchronize()

    def failed(self) -> bool:
        return bool(int(self.host[0]))


USER_MODES = ("routed", "partitioned", "replicated")


class ShardedTrainer:
    """Row-sharded tables + the training step described in the module docstring.  ``num_hashes > 0``: hash / bloom
    towers -- ``num_users`` / ``num_items`` are then BUCKET counts and ids may be arbitrary int64.

    ``user_mode``: "routed" (default) -- the user table is dealt round-robin like the item table and a rank's batch may
    hold any users (example-sharded data, as the reference's loader deals it, xfmr_rec/data/lightning.py:109): user and
    item rows travel in ONE fused exchange each way.  "partitioned" -- contiguous user blocks, every rank's batch holds
    only users of its own block (``data.DeviceInteractionSampler(user_range=...)``), user rows never travel; a user id
    outside the block raises (at the plan's host read when the plan is exact, one or two steps later otherwise).
    "replicated" -- the split ``north_star`` words: the ITEM corpus row-sharded, the USER table replicated on every rank
    (83 MB at ML-25M / d = 128); a rank's batch may hold any users, their rows are read locally (the user half of the forward
    exchange disappears), and the user-row gradients are combined by a sparse all-gather of ``(row, gradient)`` -- B x (8 + 4 d)
    bytes per rank, the row-sparse form of the all-reduce of user gradients -- after which EVERY rank applies the
    same deterministic update (duplicates summed in (rank, batch) order), so the replicas stay bit-identical.  Not for hashed
    towers (their bucket tables are sharded like the items).
    ``capacity_factor``: per-peer slots of an un-prefetched (capacity-padded) exchange as a multiple of the even share
    ``len(ids) / world``; ``None`` = ``len(ids)`` slots per peer, which can never overflow.  An overflow raises (late, like
    the id check) and leaves the step's updates incomplete -- prefetch plans, or keep ``None``."""

    def __init__(self, mf, device, optimizer: str, num_negatives: int, *, num_users: int, num_items: int, dim: int,
                 logq: torch.Tensor | None = None, kind: str = "InfomationNoiseContrastiveEstimationLoss",
                 lr: float | None = None, ops=None, comm=None, seed: int = 0, num_hashes: int = 0, hash_seed: int = 0,
                 user_mode: str = "routed", capacity_factor: float | None = None) -> None:
        if user_mode not in USER_MODES:
            msg = f"user_mode must be one of {USER_MODES}: {user_mode = }"
            raise ValueError(msg)
        self.ops = ops if ops is not None else HipOps(mf)
        self.comm = comm if comm is not None else default_comm(device)
        # the plan stream's own communicator (RcclComm: a second ncclComm; torch's group: the same object)
        self.plan_comm = self.comm.for_second_stream() if hasattr(self.comm, "for_second_stream") else self.comm
        self.rank, self.world = self.comm.rank, self.comm.world
        self.optimizer, self.num_negatives, self.kind = optimizer, num_negatives, kind
        self.hyper = optimizer_hyper(optimizer, lr)
        self.num_users, self.num_items, self.dim = num_users, num_items, dim
        self.num_hashes, self.hash_seed = int(num_hashes), int(hash_seed)
        if self.num_hashes and user_mode == "replicated":
            raise ValueError("user_mode='replicated' is for plain tables (hashed towers shard their bucket tables)")
        self.user_mode = "routed" if self.num_hashes else user_mode
        self.capacity_factor = capacity_factor
        std = 1.0 / (dim * max(self.num_hashes, 1)) ** 0.5
        # every shard is generated in place (seed + 1: the item table's stream)
        if self.user_mode == "routed":                            # dealt round-robin (bucket tables of hashed towers too)
            self.user_lo, self.user_hi = 0, num_users
            self.user_table = self.ops.init_rows(cyclic_rows(num_users, self.world, self.rank), dim, self.rank, self.world, seed,
                                                 std, device)
        elif self.user_mode == "replicated":                      # every rank generates the WHOLE user table (same counter-based values)
            self.user_lo, self.user_hi = 0, num_users
            self.user_table = self.ops.init_rows(num_users, dim, 0, 1, seed, std, device)
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
        self._checks: list[_LazyCheck] = []
        self.padded_steps = 0           # steps that ran on capacity-padded exchanges (no prefetched plan)

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
        """Unit-norm vectors of this rank's OWN user rows picked by ``rows`` (any int64: wrapped into the shard) -- query
        vectors for the retrieval leg of the benchmark."""
        local = torch.remainder(rows, max(self.user_table.shape[0], 1))
        return self.ops.gather(self.user_table, local, True)

    # -- deferred error flags -------------------------------------------------------------------
    def _raise_failed(self, block: bool) -> None:
        keep = []
        for chk in self._checks:
            if block:
                chk.wait()
            if not chk.done():
                keep.append(chk)
            elif chk.failed():
                self._checks = []
                raise _lib.MfHipError(f"step {chk.step}: {chk.message}")
        self._checks = keep

    def finish(self) -> None:
        """Wait for the error flags of the steps issued so far (id range, exchange capacity) and raise if one is set."""
        self._raise_failed(block=True)

    # -- exchange plans ------------------------------------------------------------------------
    @staticmethod
    def _key(b) -> tuple:
        return (b["item"].data_ptr(), b["item"]._version, b["item"].numel(), b["user"].data_ptr(), b["user"]._version)

    def _capacity(self, n: int) -> int:
        if self.capacity_factor is None:
            return n
        return min(n, -(-int(math.ceil(self.capacity_factor * n / self.world)) // 8) * 8)       # (whole 64-byte id lines)

    def _build(self, b, padded: bool, comm=None) -> tuple:
        """(item exchange, user exchange or None, bucket rows or None, out-of-shard flag or None)"""
        comm = comm if comm is not None else self.comm
        cap = (lambda n: self._capacity(n)) if padded else (lambda n: None)   # noqa: E731
        if self.num_hashes:
            ib = self.ops.hash_buckets(b["item"], self.num_hashes, self.hash_seed + 1, self.num_items)
            ub = self.ops.hash_buckets(b["user"], self.num_hashes, self.hash_seed, self.num_users)
            return (RowExchange(ib, comm, self.ops, cap(ib.numel())), RowExchange(ub, comm, self.ops, cap(ub.numel())),
                    (ub, ib), None)
        if self.user_mode == "routed":
            # ONE exchange for both tables: the users' ids then the items', the table in the payload's lowest bit
            ids = torch.cat([b["user"], b["item"]])
            tag = torch.zeros_like(ids)
            tag[b["user"].numel():] = 1
            ex = RowExchange(ids, comm, self.ops, cap(ids.numel()), payload=ids * 2 + tag)
            # what this rank has to gather / update for the requests it received, per table (rows of the other table are
            # asked for as -1: a zero row for the gather, skipped by the update) -- part of the PLAN: a dozen tiny
            # launches that used to sit on the step's own stream between the updates and the next forward
            req = ex.requests
            none = torch.full_like(req, -1)
            odd = torch.remainder(req, 2) == 1
            lid = torch.div(req, 2 * self.world, rounding_mode="floor")           # (id * 2 + tag) // 2 // world
            ex.table_ids = (torch.where((req >= 0) & ~odd, lid, none), torch.where((req >= 0) & odd, lid, none))
            return ex, None, None, None
        user = b["user"]
        bad = ((user < self.user_lo) | (user >= self.user_hi)).any()
        return RowExchange(b["item"], comm, self.ops, cap(b["item"].numel())), None, None, bad

    def _new_plan(self, key, b, padded: bool, ready=None, comm=None) -> _Plan:
        item, user, buckets, bad = self._build(b, padded, comm)
        if bad is not None and not padded:
            # an exact plan has just read its split sizes on the host: one more flag, on the same (plan) stream, costs
            # nothing -- and the error surfaces before the step that would have gathered zero rows
            if bool(bad):
                msg = (f"user ids outside this rank's shard [{self.user_lo}, {self.user_hi}) in a user-partitioned batch "
                       "(user_mode='partitioned' needs per-rank batches of its own users; use user_mode='routed' for "
                       "example-sharded data)" if self.user_mode == "partitioned" else
                       f"user ids outside the user table [0, {self.num_users})")
                raise _lib.MfHipError(msg)
            bad = None
        return _Plan(key, item, user, ready, self.steps, buckets, bad)

    def _plan(self, b) -> _Plan:
        """The routing plan of a batch.  An exact plan costs a host sync (the split sizes); ``prefetch`` moves that sync
        off the critical path.  A prefetched plan is only used for the very tensors it was built from (same storage, same
        version: the plan holds a reference, so the allocator cannot hand that address to another batch meanwhile).
        Without one the step runs on capacity-padded exchanges: no host sync at all."""
        key = self._key(b)
        hit = self._plans.pop(key, None)
        # whatever else was prefetched and not consumed by now (epoch end, skipped batch) is stale: drop it
        for k in [k for k, p in self._plans.items() if p.built_at < self.steps - 1]:
            del self._plans[k]
        if hit is None:
            self.padded_steps += 1
            return self._new_plan(key, b, padded=True)
        for ex in (hit.item, hit.user):            # built with the plan stream's communicator: the step's exchanges use its own
            if ex is not None:
                ex.comm = self.comm
        if hit.ready is not None:                  # built on the side stream: order it before our use
            cur = torch.cuda.current_stream()
            cur.wait_event(hit.ready)
            for ex in (hit.item, hit.user):
                if ex is not None:
                    for t in (ex.order, ex.local_ids, ex.requests) + tuple(getattr(ex, "table_ids", ())):
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
            self._plans[key] = self._new_plan(key, next_b, padded=False, comm=self.plan_comm)
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
            plan = self._new_plan(key, next_b, padded=False, comm=self.plan_comm)
            plan.ready = torch.cuda.Event()
            plan.ready.record(self._plan_stream)
        for t in (next_b["item"], next_b["user"]):
            t.record_stream(self._plan_stream)
        self._plans[key] = plan

    # -- one step -----------------------------------------------------------------------------
    def step(self, b, next_b=None) -> torch.Tensor:
        """``b``: ``user`` (global user rows -- any, or this rank's block in "partitioned" mode; any ids with hashed
        towers), ``item`` (2B global rows: positives then negatives), ``target``, ``pos`` (padded positives) or ``pos_csr``
        ((user ids, pos_off, pos_items): the lists in place).  ``next_b``: the batch
        after it, if known (its exchange plan is then prefetched behind this step's compute)."""
        self.steps += 1
        self._raise_failed(block=False)           # flags of earlier steps whose host copies have landed by now
        entry = None
        if next_b is not None and b["item"].is_cuda:
            entry = torch.cuda.Event()
            entry.record()                        # next_b's ids are queued by now; this step's sweeps are not
        plan = self._plan(b)
        ops, H = self.ops, self.num_hashes
        for ex in (plan.item, plan.user):
            if ex is not None and ex.overflow is not None and self.capacity_factor is not None:
                self._checks.append(_LazyCheck(self.steps, f"exchange capacity exceeded (capacity_factor = {self.capacity_factor}); "
                                               "this step's updates are incomplete", ex.overflow))
        if plan.bad_users is not None:            # (padded plan: nothing is read back now; an exact plan checked at build time)
            self._checks.append(_LazyCheck(self.steps, (f"user ids outside this rank's shard [{self.user_lo}, {self.user_hi}) in a "
                                           "user-partitioned batch (use user_mode='routed' for example-sharded data)" if self.user_mode == "partitioned"
                                           else f"user ids outside the user table [0, {self.num_users})") + "; the rows "
                                           "were gathered as zeros and not updated", plan.bad_users))
        nb = b["user"].numel()
        if H:
            v, v_inv = ops.bloom_forward(plan.item.fetch(ops.gather(self.item_table, plan.item.local_ids, False)), H)
            u, u_inv = ops.bloom_forward(plan.user.fetch(ops.gather(self.user_table, plan.user.local_ids, False)), H)
        elif self.user_mode == "routed":
            # requests carry the table in bit 0: rows of the other table are asked for as -1 (a zero row), so the two
            # gathers add up to the mixed block exactly
            uid, iid = plan.item.table_ids
            rows = plan.item.fetch(ops.gather(self.user_table, uid, True) + ops.gather(self.item_table, iid, True))
            u, v = rows[:nb], rows[nb:]
        else:                                     # "partitioned" / "replicated": the users' rows are here
            user_local = b["user"] - self.user_lo
            v = plan.item.fetch(ops.gather(self.item_table, plan.item.local_ids, True))
            u = ops.gather(self.user_table, user_local, True)
        kw = {"pos_csr": b["pos_csr"]} if b.get("pos_csr") is not None else {}      # (CSR lists: HipOps only)
        loss, du, dv = ops.loss_and_grads(self.kind, u, v, b["target"], b["item"], b.get("pos"), self.logq, self.num_negatives, 1.0, 1.0, **kw)
        if next_b is not None:
            self.prefetch(next_b, after=entry)    # the GPU is busy with the sweeps just queued
        if H:
            dv_owned = plan.item.push(ops.bloom_backward(v, v_inv, dv, H))
            du_owned = plan.user.push(ops.bloom_backward(u, u_inv, du, H))
            ops.update(self.optimizer, self.item_table, self.state["item"], plan.item.local_ids, dv_owned, False, self.steps, self.hyper)
            ops.update(self.optimizer, self.user_table, self.state["user"], plan.user.local_ids, du_owned, False, self.steps, self.hyper)
        elif self.user_mode == "routed":
            owned = plan.item.push(torch.cat([du, dv]))
            ops.update(self.optimizer, self.item_table, self.state["item"], iid, owned, True, self.steps, self.hyper)
            ops.update(self.optimizer, self.user_table, self.state["user"], uid, owned, True, self.steps, self.hyper)
        elif self.user_mode == "replicated":
            dv_owned = plan.item.push(dv)
            ops.update(self.optimizer, self.item_table, self.state["item"], plan.item.local_ids, dv_owned, True, self.steps, self.hyper)
            # the sparse form of "all-reduce of user gradients": every rank receives every rank's (row, gradient) pairs, in rank
            # order, and applies ONE update to its replica -- the same list, the same deterministic kernel: identical replicas
            ids_all, du_all = self.comm.gather(b["user"]), self.comm.gather(du)
            ops.update(self.optimizer, self.user_table, self.state["user"], ids_all, du_all, True, self.steps, self.hyper)
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

    def search(self, queries: torch.Tensor, top_k: int, *, exclude_csr=None, exclude_capacity: int | None = None):
        """``exclude_csr = (off [q + 1], ids)``: this rank's queries' exclusion lists (global item rows).  Every rank needs every
        rank's lists (it scans its shard for ALL queries).  Default: the lists travel at their exact lengths, which costs one
        host read of the other ranks' lengths per NEW set of lists.  ``exclude_capacity = C`` (a bound on ``ids.numel()`` the
        caller knows on the host, the same on every rank): every rank sends a block of C entries padded with -1 and the
        offsets are rebased onto the blocks on the device -- no host sync at all (the padding lands in the last query of each
        block as ids no shard owns, which the scan ignores)."""
        comm = self.comm
        world = comm.world
        q, d = queries.shape
        all_q = comm.gather(queries)
        csr = None
        key = None if exclude_csr is None else (exclude_csr[0].data_ptr(), exclude_csr[1].data_ptr(), exclude_csr[1].numel(), q, exclude_capacity)
        if key is not None and getattr(self, "_csr_key", None) == key:
            csr = self._csr_all                       # same exclusion lists as the last call: gathered once
        elif exclude_csr is not None and exclude_capacity is not None:
            off, ids = exclude_csr
            cap = int(exclude_capacity)
            if ids.numel() > cap or cap < 1:
                msg = f"exclude_capacity = {cap} but this rank's lists hold {ids.numel()} entries"
                raise ValueError(msg)
            padded = torch.full((cap,), -1, dtype=torch.int64, device=queries.device)
            padded[: ids.numel()] = ids
            all_ids = comm.gather(padded)                                        # [world * cap]
            all_off = comm.gather(off.contiguous()).reshape(world, q + 1)        # every rank's offsets, from 0
            base = torch.arange(world, device=queries.device, dtype=torch.int64)[:, None] * cap
            starts = (all_off[:, :q] + base).reshape(-1)                         # query (r, i) starts inside block r ...
            end = torch.full((1,), world * cap, dtype=torch.int64, device=queries.device)
            csr = (torch.cat([starts, end]), self._localise(all_ids))            # ... and a block's padding rides with its last query: -1, ignored
            self._csr_key, self._csr_all, self._csr_src = key, csr, exclude_csr
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
