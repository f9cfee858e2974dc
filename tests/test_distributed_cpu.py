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

    def gather(self, table, ids, normalize):
        return oembed.gather(table, ids, normalize)

    def loss_and_grads(self, kind, u, v, target, item_idx, pos_idx, logq, num_negatives, sigma, margin):
        u = u.detach().clone().requires_grad_()
        v = v.detach().clone().requires_grad_()
        loss = ol.loss(kind, u, v, target, item_idx=item_idx, pos_idx=pos_idx, num_negatives=num_negatives, sigma=sigma,
                       margin=margin, logq=logq)
        loss.backward()
        return loss.detach(), u.grad, v.grad

    def update(self, optimizer, table, state, ids, grad, normalized, step, lr):
        if ids.numel() == 0:
            return
        g = oembed.normalize_backward(table[ids], grad) if normalized else grad
        if optimizer == "sgd":
            oembed.sgd_update(table, ids, g, lr)
        else:
            oembed.adam_update(table, state["m"], state["v"], ids, g, step=step, lr=lr, weight_decay=0.01)

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


def _batch(rank: int, world: int, mfd):
    g = torch.Generator().manual_seed(100 + rank)
    lo, hi = mfd.shard_bounds(N_USERS, world, rank)
    item = torch.randint(0, N_ITEMS, (2 * B,), generator=g)
    item[:4] = 3                                       # duplicates, all owned by one rank
    pos = torch.randint(0, N_ITEMS, (B, P), generator=g)
    pos[:, 0] = item[:B]
    return {"user": torch.randint(lo, hi, (B,), generator=g), "item": item,
            "target": torch.randint(1, 6, (B,), generator=g), "pos": pos}


def _train_case(rank: int, world: int, out_dir: str) -> None:
    mf = importlib.import_module("matrix-factorization-torch_amd")
    mfd = mf.distributed
    for opt in ("sgd", "adam"):
        tr = mfd.ShardedTrainer(mf, "cpu", opt, 0, num_users=N_USERS, num_items=N_ITEMS, dim=DIM, ops=OracleOps(), lr=0.05,
                                kind="PairwiseLogisticLoss")
        b = _batch(rank, world, mfd)
        tr.prefetch(b)                                   # the plan built ahead of time is the one the step uses
        assert b["item"].data_ptr() in tr._plans
        loss = tr.step(b, next_b=b)
        assert list(tr._plans) == [b["item"].data_ptr()]  # consumed, and the next one prefetched
        torch.save({"user": tr.user_table, "item": tr.item_table, "loss": loss}, f"{out_dir}/{opt}_{rank}.pt")


def test_sharded_training_step_matches_single_process(tmp_path):
    """After one step the concatenated shards equal one process applying every rank's gradients
    (computed from the same pre-step tables) in a single sparse update per table."""
    mf = importlib.import_module("matrix-factorization-torch_amd")
    mfd = mf.distributed
    _run("_train_case", tmp_path)
    for opt in ("sgd", "adam"):
        g = torch.Generator().manual_seed(0)
        ut = torch.randn(N_USERS, DIM, generator=g) / DIM**0.5
        it = torch.randn(N_ITEMS, DIM, generator=g) / DIM**0.5
        ut0, it0 = ut.clone(), it.clone()
        ops = OracleOps()
        u_ids, u_g, i_ids, i_g, losses = [], [], [], [], []
        for r in range(2):
            b = _batch(r, 2, mfd)
            loss, du, dv = ops.loss_and_grads("PairwiseLogisticLoss", oembed.gather(ut0, b["user"], True),
                                              oembed.gather(it0, b["item"], True), b["target"], b["item"], b["pos"], None,
                                              0, 1.0, 1.0)
            u_ids.append(b["user"]); u_g.append(du); i_ids.append(b["item"]); i_g.append(dv); losses.append(loss)
        st = {"m": torch.zeros_like(ut), "v": torch.zeros_like(ut)}
        ops.update(opt, ut, st, torch.cat(u_ids), torch.cat(u_g), True, 1, 0.05)
        st = {"m": torch.zeros_like(it), "v": torch.zeros_like(it)}
        ops.update(opt, it, st, torch.cat(i_ids), torch.cat(i_g), True, 1, 0.05)
        got = [torch.load(f"{tmp_path}/{opt}_{r}.pt") for r in range(2)]
        torch.testing.assert_close(torch.cat([x["user"] for x in got]), ut, rtol=1e-5, atol=1e-6)
        items = torch.empty_like(it)
        for r in range(2):
            items[r::2] = got[r]["item"]              # item rows are dealt round-robin
        torch.testing.assert_close(items, it, rtol=1e-5, atol=1e-6)
        for r in range(2):
            torch.testing.assert_close(got[r]["loss"], losses[r])


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
