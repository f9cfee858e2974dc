"""CPU, world_size 2, gloo: the sharding / exchange logic of distributed.py with the oracle
injected as the local compute (the HIP kernels themselves are covered by -m gpu)."""
from __future__ import annotations

import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import chain, embed as oembed, losses as ol, retrieval as oretr


class OracleOps:
    """Local compute by the CPU oracle (test infrastructure standing in for HipOps)."""

    def init_rows(self, n_local, d, row_start, row_stride, seed, std, device):
        return oembed.init_rows(n_local, d, row_start, row_stride, seed, std)

    def gather(self, table, ids, normalize, want_inv=False):
        # ids outside the table (the -1 of a padded or other-table request) give a zero row, like mf_gather_rows
        ok = (ids >= 0) & (ids < table.shape[0])
        rows = table[torch.where(ok, ids, torch.zeros_like(ids))] * ok[:, None].to(table.dtype)
        if not normalize:
            return (rows, torch.ones(ids.numel())) if want_inv else rows
        inv = 1.0 / rows.norm(dim=-1).clamp_min(1e-12)
        return (rows * inv[:, None], inv) if want_inv else rows * inv[:, None]

    def hash_buckets(self, ids, num_hashes, seed, num_buckets):
        return oembed.hash_buckets(ids, num_hashes, seed, num_buckets).reshape(-1)

    def bloom_forward(self, bucket_rows, num_hashes):
        n = bucket_rows.shape[0] // num_hashes
        raw = bucket_rows.view(n, num_hashes, -1).sum(dim=1)
        inv = 1.0 / raw.norm(dim=-1).clamp_min(1e-12)
        return raw * inv[:, None], inv

    def bloom_backward(self, unit, inv, grad, num_hashes):
        graw = (grad - unit * (grad * unit).sum(-1, keepdim=True)) * inv[:, None]
        return graw.repeat_interleave(num_hashes, dim=0)

    def loss_and_grads(self, kind, u, v, target, item_idx, pos_idx, logq_table, num_negatives, sigma, margin):
        u = u.detach().clone().requires_grad_()
        v = v.detach().clone().requires_grad_()
        loss = ol.loss(kind, u, v, target, item_idx=item_idx, pos_idx=pos_idx, num_negatives=num_negatives, sigma=sigma,
                       margin=margin, logq=None if logq_table is None else logq_table[item_idx])
        loss.backward()
        return loss.detach(), u.grad, v.grad

    def update(self, optimizer, table, state, ids, grad, normalized, step, hyper):
        keep = (ids >= 0) & (ids < table.shape[0])          # out-of-range ids are skipped, like mf_update_*
        ids, grad = ids[keep], grad[keep]
        if ids.numel() == 0:
            return
        g = oembed.normalize_backward(table[ids], grad) if normalized else grad
        if optimizer == "sgd":
            oembed.sgd_update(table, ids, g, hyper["lr"], hyper["weight_decay"])
        else:
            oembed.adam_update(table, state["m"], state["v"], ids, g, step=step, lr=hyper["lr"], beta1=hyper["betas"][0],
                               beta2=hyper["betas"][1], eps=hyper["eps"], weight_decay=hyper["weight_decay"])

    def topk(self, queries, items, k, exclude_csr, idx_base):
        excl = None
        if exclude_csr is not None:
            off, ids = exclude_csr
            n = items.shape[0]
            excl = [[int(i) - idx_base for i in ids[off[r]:off[r + 1]].tolist() if 0 <= int(i) - idx_base < n]
                    for r in range(queries.shape[0])]
        s, i = chain.topk(queries.numpy(), items.numpy(), k, excl)
        i = np.where(i >= 0, i + idx_base, -1)
        return torch.from_numpy(s), torch.from_numpy(i)

    def merge(self, ps, pi, k):
        s, i = oretr.merge_topk(ps.numpy(), pi.numpy(), k)
        return torch.from_numpy(s), torch.from_numpy(i)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, fn: str, out_dir: str) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        globals()[fn](rank, world, out_dir)
    finally:
        dist.destroy_process_group()


def _run(fn: str, tmp_path, world: int = 2) -> None:
    torch.set_num_threads(1)
    mp.spawn(_worker, args=(world, _free_port(), fn, str(tmp_path)), nprocs=world, join=True)


N_USERS, N_ITEMS, DIM, B, P, K = 64, 101, 32, 16, 4, 7


def _batch(rank: int, world: int, mfd, mode: str = "routed", salt: int = 0):
    """"routed": example-sharded, as the reference's loader deals examples (data/lightning.py:109) -- any user on any rank,
    the same user on several ranks; "partitioned": every rank draws from its own user block."""
    g = torch.Generator().manual_seed(100 + rank + 1000 * salt)
    lo, hi = mfd.shard_bounds(N_USERS, world, rank) if mode == "partitioned" else (0, N_USERS)      # ("replicated": any user, like "routed")
    item = torch.randint(0, N_ITEMS, (2 * B,), generator=g)
    item[:4] = 3                                       # duplicates, all owned by one rank
    pos = torch.randint(0, N_ITEMS, (B, P), generator=g)
    pos[:, 0] = item[:B]
    user = torch.randint(lo, hi, (B,), generator=g)
    if mode in ("routed", "replicated"):
        user[:3] = 5                                   # one user three times here -- and on every other rank too
    return {"user": user, "item": item, "target": torch.randint(1, 6, (B,), generator=g), "pos": pos}


def _save(tr, loss, out_dir, tag, rank):
    torch.save({"user": tr.user_table, "item": tr.item_table, "loss": loss}, f"{out_dir}/{tag}_{rank}.pt")


def _train_case(rank: int, world: int, out_dir: str, mode: str = "routed") -> None:
    mf = importlib.import_module("matrix-factorization-torch_amd")
    mfd = mf.distributed
    for opt in ("sgd", "adam"):
        def trainer(**kw):
            return mfd.ShardedTrainer(mf, "cpu", opt, 0, num_users=N_USERS, num_items=N_ITEMS, dim=DIM, ops=OracleOps(), lr=0.05,
                                      kind="PairwiseLogisticLoss", user_mode=mode, **kw)

        tr = trainer()
        assert isinstance(tr.comm, mfd.TorchComm)
        b = _batch(rank, world, mfd, mode)
        tr.prefetch(b)                                   # the plan built ahead of time (exact counts) is the one the step uses
        assert tr._key(b) in tr._plans and tr._plans[tr._key(b)].item.capacity is None
        # a plan is bound to the very tensors it was built from: a batch that merely looks alike gets its own
        other = {k: v.clone() for k, v in b.items()}
        assert tr._key(other) not in tr._plans
        loss = tr.step(b, next_b=b)
        assert list(tr._plans) == [tr._key(b)] and tr.padded_steps == 0    # consumed, and the next one prefetched
        _save(tr, loss, out_dir, f"{opt}_exact", rank)
        # a prefetched plan that is never consumed (skipped batch, epoch end) is dropped, not matched to a later batch
        tr.prefetch(other)
        for j in range(3):
            tr.step(_batch(rank, world, mfd, mode, salt=j + 1))
        assert tr._key(other) not in tr._plans and tr.padded_steps == 3
        tr.finish()
        # the same first step WITHOUT a prefetched plan: capacity-padded exchanges (no host read), same result
        tr2 = trainer()
        loss2 = tr2.step(b)
        assert tr2.padded_steps == 1
        tr2.finish()
        _save(tr2, loss2, out_dir, f"{opt}_padded", rank)
        # ... and with a capacity that holds (2 x the even share, rounded up to 64 slots)
        tr3 = trainer(capacity_factor=2.0)
        loss3 = tr3.step(b)
        tr3.finish()
        _save(tr3, loss3, out_dir, f"{opt}_cap2", rank)


def _train_case_partitioned(rank: int, world: int, out_dir: str) -> None:
    _train_case(rank, world, out_dir, mode="partitioned")


def _train_case_replicated(rank: int, world: int, out_dir: str) -> None:
    _train_case(rank, world, out_dir, mode="replicated")


def _check_train(tmp_path, world: int, mode: str = "routed", tags=("exact", "padded", "cap2"), batch_fn=None, rtol=1e-5, atol=1e-6) -> None:
    """The shards of every rank, put back together, against ONE process applying every rank's gradients (computed from the
    same pre-step tables) in a single sparse update per table."""
    mf = importlib.import_module("matrix-factorization-torch_amd")
    mfd = mf.distributed
    std = 1.0 / DIM**0.5
    batch_fn = batch_fn or (lambda r: _batch(r, world, mfd, mode))
    for opt in ("sgd", "adam"):
        ut = oembed.init_rows(N_USERS, DIM, 0, 1, 0, std)          # the virtual tables every shard was cut from
        it = oembed.init_rows(N_ITEMS, DIM, 0, 1, 1, std)
        ut0, it0 = ut.clone(), it.clone()
        ops = OracleOps()
        hyper = mfd.optimizer_hyper(opt, 0.05)
        u_ids, u_g, i_ids, i_g, losses = [], [], [], [], []
        for r in range(world):
            b = batch_fn(r)
            loss, du, dv = ops.loss_and_grads("PairwiseLogisticLoss", oembed.gather(ut0, b["user"], True),
                                              oembed.gather(it0, b["item"], True), b["target"], b["item"], b["pos"], None,
                                              0, 1.0, 1.0)
            u_ids.append(b["user"]); u_g.append(du); i_ids.append(b["item"]); i_g.append(dv); losses.append(loss)
        st = {"m": torch.zeros_like(ut), "v": torch.zeros_like(ut)}
        ops.update(opt, ut, st, torch.cat(u_ids), torch.cat(u_g), True, 1, hyper)
        st = {"m": torch.zeros_like(it), "v": torch.zeros_like(it)}
        ops.update(opt, it, st, torch.cat(i_ids), torch.cat(i_g), True, 1, hyper)
        for tag in tags:
            got = [torch.load(f"{tmp_path}/{opt}_{tag}_{r}.pt") for r in range(world)]
            if mode == "replicated":                          # every rank holds the whole user table: identical replicas
                users = got[0]["user"]
                for r in range(1, world):
                    assert torch.equal(got[r]["user"], users), (opt, tag, r)
            elif mode == "partitioned":                       # contiguous user blocks
                users = torch.cat([x["user"] for x in got])
            else:                                             # user rows dealt round-robin, like the items
                users = torch.empty_like(ut)
                for r in range(world):
                    users[r::world] = got[r]["user"]
            torch.testing.assert_close(users, ut, rtol=rtol, atol=atol)
            items = torch.empty_like(it)
            for r in range(world):
                items[r::world] = got[r]["item"]              # item rows are dealt round-robin
            torch.testing.assert_close(items, it, rtol=rtol, atol=atol)
            for r in range(world):
                torch.testing.assert_close(got[r]["loss"], losses[r], rtol=max(rtol, 1.3e-6), atol=max(atol, 1e-5))


def test_sharded_training_step_matches_single_process(tmp_path):
    """EXAMPLE-sharded batches (any user on any rank, as the reference's loader deals them): after one step the shards equal
    one process applying every rank's gradients in a single sparse update per table -- through a prefetched exact plan,
    through capacity-padded exchanges (no host sync) and through a tighter capacity."""
    _run("_train_case", tmp_path)
    _check_train(tmp_path, 2)


def test_sharded_training_step_world_4(tmp_path):
    _run("_train_case", tmp_path, world=4)
    _check_train(tmp_path, 4)


def test_sharded_training_step_user_partitioned(tmp_path):
    """The fast path of a user-partitioned stream: user rows never travel."""
    _run("_train_case_partitioned", tmp_path)
    _check_train(tmp_path, 2, mode="partitioned")


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_training_step_user_replicated(tmp_path, world):
    """The split ``north_star`` words: item corpus row-sharded, user table REPLICATED, user-row gradients combined by a
    sparse all-gather of (row, gradient) that every rank applies -- equal to one process applying every rank's gradients,
    and the replicas bit-identical to each other (example-sharded batches; exact, padded and tight-capacity item plans)."""
    _run("_train_case_replicated", tmp_path, world=world)
    _check_train(tmp_path, world, mode="replicated")


class _TaggedComm:
    """Wraps a communicator and logs which instance served which call."""

    def __init__(self, inner, tag, log):
        self.inner, self.tag, self.log = inner, tag, log
        self.world, self.rank, self.transport, self.rccl_ranks = inner.world, inner.rank, inner.transport, None

    def for_second_stream(self):
        return _TaggedComm(self.inner, "plan", self.log)

    def __getattr__(self, name):
        fn = getattr(self.inner, name)

        def call(*a, **k):
            self.log.append((self.tag, name))
            return fn(*a, **k)

        return call


def _two_comm_case(rank: int, world: int, out_dir: str) -> None:
    mf = importlib.import_module("matrix-factorization-torch_amd")
    mfd = mf.distributed
    log = []
    tr = mfd.ShardedTrainer(mf, "cpu", "sgd", 0, num_users=N_USERS, num_items=N_ITEMS, dim=DIM, ops=OracleOps(), lr=0.05,
                            kind="PairwiseLogisticLoss", comm=_TaggedComm(mfd.TorchComm(), "step", log))
    assert tr.plan_comm is not tr.comm and tr.plan_comm.tag == "plan"
    b = _batch(rank, world, mfd)
    tr.prefetch(b)
    assert log and all(tag == "plan" for tag, _ in log), log          # the plan's exchanges: the second communicator only
    n_plan = len(log)
    tr.step(b)
    assert len(log) > n_plan and all(tag == "step" for tag, _ in log[n_plan:]), log[n_plan:]      # fetch / push: the step's own
    tr.finish()


def test_prefetched_plans_use_their_own_communicator(tmp_path):
    """One ncclComm must not be driven from two streams: a plan prefetched on the side stream exchanges through
    ``comm.for_second_stream()`` (RcclComm: a second communicator), the step's fetch / push through the step's."""
    _run("_two_comm_case", tmp_path)


def _guard_case(rank: int, world: int, out_dir: str) -> None:
    """Errors instead of silent zero rows / silently dropped rows."""
    mf = importlib.import_module("matrix-factorization-torch_amd")
    mfd = mf.distributed
    kw = dict(num_users=N_USERS, num_items=N_ITEMS, dim=DIM, ops=OracleOps(), lr=0.05, kind="PairwiseLogisticLoss")
    # user-partitioned trainer fed an example-sharded batch: raises when the exact plan is built ...
    tr = mfd.ShardedTrainer(mf, "cpu", "sgd", 0, user_mode="partitioned", **kw)
    b = _batch(rank, world, mfd, "routed")
    b["user"][0] = (tr.user_hi + 1) % N_USERS if world > 1 else b["user"][0]       # certainly another rank's row
    with pytest.raises(mf.MfHipError, match="outside this rank's shard"):
        tr.prefetch(b)
    # (every rank raised at the same point: the collectives stay aligned) ... and after a padded step, at finish()
    tr = mfd.ShardedTrainer(mf, "cpu", "sgd", 0, user_mode="partitioned", **kw)
    tr.step(b)
    with pytest.raises(mf.MfHipError, match="outside this rank's shard"):
        tr.finish()
    # a capacity too small for a batch whose items all live on one owner
    tr = mfd.ShardedTrainer(mf, "cpu", "sgd", 0, capacity_factor=0.5, **kw)
    b = _batch(rank, world, mfd, "routed")
    b["item"][:] = 3
    b["user"][:] = 3
    tr.step(b)
    with pytest.raises(mf.MfHipError, match="capacity exceeded"):
        tr.finish()
    with pytest.raises(ValueError, match="user_mode"):
        mfd.ShardedTrainer(mf, "cpu", "sgd", 0, user_mode="mirrored", **kw)
    with pytest.raises(ValueError, match="replicated"):
        mfd.ShardedTrainer(mf, "cpu", "sgd", 0, user_mode="replicated", num_hashes=2, **kw)


def test_sharded_trainer_raises_instead_of_zero_rows(tmp_path):
    _run("_guard_case", tmp_path)


# hash / bloom towers (BASELINE config 5 shape: ids far beyond the table height, d = 256 there; 2 hashes)
HB_U, HB_I, HD, HH = 23, 57, 32, 2


def _hashed_batch(rank: int):
    g = torch.Generator().manual_seed(300 + rank)
    item = torch.randint(0, 100_000_000, (2 * B,), generator=g)
    item[B: B + 3] = item[:3]                         # negatives colliding with positives
    pos = torch.randint(0, 100_000_000, (B, P), generator=g)
    pos[:, 0] = item[:B]
    return {"user": torch.randint(0, 10_000_000, (B,), generator=g), "item": item,
            "target": torch.randint(1, 6, (B,), generator=g), "pos": pos}


def _hashed_case(rank: int, world: int, out_dir: str) -> None:
    mf = importlib.import_module("matrix-factorization-torch_amd")
    tr = mf.distributed.ShardedTrainer(mf, "cpu", "sgd", 0, num_users=HB_U, num_items=HB_I, dim=HD, ops=OracleOps(), lr=0.05,
                                       num_hashes=HH, hash_seed=3)
    b = _hashed_batch(rank)
    loss = tr.step(b, next_b=b)
    tr.finish()
    torch.save({"user": tr.user_table, "item": tr.item_table, "loss": loss}, f"{out_dir}/hashed_{rank}.pt")


def test_sharded_hashed_towers_match_single_process(tmp_path):
    """Config 5's structure on two ranks: both bucket tables dealt round-robin, every id's bucket rows fetched from
    their owners, summed, normalised; bucket-row gradients pushed back and applied once per owner -- equal to one process
    doing the same step on the whole bucket tables."""
    world = 2
    _run("_hashed_case", tmp_path, world=world)
    std = 1.0 / (HD * HH) ** 0.5
    tu, ti = oembed.init_rows(HB_U, HD, 0, 1, 0, std), oembed.init_rows(HB_I, HD, 0, 1, 1, std)
    tu0, ti0 = tu.clone(), ti.clone()
    ub_all, ug_all, ib_all, ig_all, losses = [], [], [], [], []
    for r in range(world):
        b = _hashed_batch(r)
        bu, bi = oembed.hash_buckets(b["user"], HH, 3, HB_U), oembed.hash_buckets(b["item"], HH, 4, HB_I)
        raw_u = (tu0[bu[:, 0]] + tu0[bu[:, 1]]).requires_grad_()
        raw_i = (ti0[bi[:, 0]] + ti0[bi[:, 1]]).requires_grad_()
        un = raw_u / raw_u.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        vn = raw_i / raw_i.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        loss = ol.loss("InfomationNoiseContrastiveEstimationLoss", un, vn, b["target"], item_idx=b["item"], pos_idx=b["pos"])
        loss.backward()
        losses.append(loss.detach())
        ub_all.append(bu.reshape(-1)); ug_all.append(raw_u.grad.repeat_interleave(HH, dim=0))
        ib_all.append(bi.reshape(-1)); ig_all.append(raw_i.grad.repeat_interleave(HH, dim=0))
    oembed.sgd_update(tu, torch.cat(ub_all), torch.cat(ug_all), 0.05, 0.0)
    oembed.sgd_update(ti, torch.cat(ib_all), torch.cat(ig_all), 0.05, 0.0)
    got = [torch.load(f"{tmp_path}/hashed_{r}.pt") for r in range(world)]
    for name, want in (("user", tu), ("item", ti)):
        full = torch.empty_like(want)
        for r in range(world):
            full[r::world] = got[r][name]
        torch.testing.assert_close(full, want, rtol=1e-5, atol=1e-6)
    for r in range(world):
        torch.testing.assert_close(got[r]["loss"], losses[r], rtol=1e-5, atol=1e-6)


def _topk_case(rank: int, world: int, out_dir: str) -> None:
    mf = importlib.import_module("matrix-factorization-torch_amd")
    mfd = mf.distributed
    g = torch.Generator().manual_seed(5)
    items = torch.nn.functional.normalize(torch.randn(N_ITEMS, DIM, generator=g), dim=-1)
    gq = torch.Generator().manual_seed(50 + rank)
    q = torch.nn.functional.normalize(torch.randn(5, DIM, generator=gq), dim=-1)
    excl = [sorted(set(torch.randint(0, N_ITEMS, (int(n),), generator=gq).tolist())) for n in (0, 3, 9, 1, 20)]
    off = torch.tensor([0] + list(np.cumsum([len(e) for e in excl])), dtype=torch.int64)
    ids = torch.tensor([i for e in excl for i in e] or [0], dtype=torch.int64)
    index = mfd.ShardedIndex(items[rank::world].contiguous(), rank, N_ITEMS, stride=world, ops=OracleOps())
    s, i = index.search(q, K, exclude_csr=(off, ids))
    # the same search with capacity-padded exclusion blocks (no host read of the other ranks' list lengths): same bits
    index2 = mfd.ShardedIndex(items[rank::world].contiguous(), rank, N_ITEMS, stride=world, ops=OracleOps())
    s2, i2 = index2.search(q, K, exclude_csr=(off, ids.clone()), exclude_capacity=40)
    assert torch.equal(i, i2) and torch.equal(s, s2)
    with pytest.raises(ValueError, match="exclude_capacity"):
        index2.search(q, K, exclude_csr=(off, ids.clone()), exclude_capacity=8)
    torch.save({"q": q, "excl": excl, "s": s, "i": i}, f"{out_dir}/topk_{rank}.pt")


def test_sharded_topk_is_bit_identical_to_full_scan(tmp_path):
    _run("_topk_case", tmp_path)
    g = torch.Generator().manual_seed(5)
    items = torch.nn.functional.normalize(torch.randn(N_ITEMS, DIM, generator=g), dim=-1)
    for r in range(2):
        got = torch.load(f"{tmp_path}/topk_{r}.pt")
        ws, wi = chain.topk(got["q"].numpy(), items.numpy(), K, got["excl"])
        assert np.array_equal(got["i"].numpy(), wi)
        assert np.array_equal(got["s"].numpy().view(np.uint32), ws.view(np.uint32))
