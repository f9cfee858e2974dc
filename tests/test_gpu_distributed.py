"""GPU, world_size 2 on ONE card: the sharded trainer and index with the HIP kernels as local compute (``HipOps``) and
the exchanges relayed through host memory over ``gloo`` -- everything of the multi-GPU path except RCCL itself, which
needs two cards (``RcclComm`` at world 1 is covered in test_gpu_module.py; at world > 1 ``default_comm`` self-tests it
against ``torch.distributed`` before trusting it)."""
from __future__ import annotations

import importlib
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import chain, embed as oembed
from tests import test_distributed_cpu as tdc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _relay_comm(mfd):
    class RelayComm(mfd.TorchComm):
        """torch.distributed (gloo) on host copies: a stand-in for the xGMI exchange on a one-GPU box."""

        def rows(self, x, send_counts, recv_counts):
            return super().rows(x.cpu(), send_counts, recv_counts).to(DEV)

        def counts(self, send):
            return super().counts(send.cpu()).to(DEV)

        def gather(self, x):
            return super().gather(x.cpu()).to(DEV)

        def equal(self, x):
            return super().equal(x.cpu()).to(DEV)

    return RelayComm()


def _gpu_train_case(rank: int, world: int, out_dir: str) -> None:
    mf = importlib.import_module("matrix-factorization-torch_amd")
    mfd = mf.distributed
    for opt in ("sgd", "adam"):
        tr = mfd.ShardedTrainer(mf, DEV, opt, 0, num_users=tdc.N_USERS, num_items=tdc.N_ITEMS, dim=tdc.DIM, lr=0.05,
                                kind="PairwiseLogisticLoss", comm=_relay_comm(mfd))
        assert isinstance(tr.ops, mfd.HipOps)
        b = {k: v.to(DEV) for k, v in tdc._batch(rank, world, mfd).items()}
        loss = tr.step(b, next_b=b)
        torch.cuda.synchronize()
        torch.save({"user": tr.user_table.cpu(), "item": tr.item_table.cpu(), "loss": loss.cpu()}, f"{out_dir}/{opt}_{rank}.pt")


def _gpu_topk_case(rank: int, world: int, out_dir: str) -> None:
    mf = importlib.import_module("matrix-factorization-torch_amd")
    mfd = mf.distributed
    g = torch.Generator().manual_seed(5)
    items = torch.nn.functional.normalize(torch.randn(tdc.N_ITEMS, tdc.DIM, generator=g), dim=-1)
    gq = torch.Generator().manual_seed(50 + rank)
    q = torch.nn.functional.normalize(torch.randn(5, tdc.DIM, generator=gq), dim=-1)
    excl = [sorted(set(torch.randint(0, tdc.N_ITEMS, (6,), generator=gq).tolist())) for _ in range(5)]
    off = torch.tensor([0] + list(np.cumsum([len(e) for e in excl])), dtype=torch.int64)
    ids = torch.tensor([i for e in excl for i in e] or [0], dtype=torch.int64)
    index = mfd.ShardedIndex(items[rank::world].contiguous().to(DEV), rank, tdc.N_ITEMS, stride=world, comm=_relay_comm(mfd))
    s, i = index.search(q.to(DEV), tdc.K, exclude_csr=(off.to(DEV), ids.to(DEV)))
    torch.save({"q": q, "excl": excl, "s": s.cpu(), "i": i.cpu()}, f"{out_dir}/topk_{rank}.pt")


def _worker(rank: int, world: int, port: int, fn: str, out_dir: str) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        globals()[fn](rank, world, out_dir)
    finally:
        dist.destroy_process_group()


def _run(fn: str, tmp_path, world: int = 2) -> None:
    mp.spawn(_worker, args=(world, tdc._free_port(), fn, str(tmp_path)), nprocs=world, join=True)


def test_sharded_step_world_2_with_hip_kernels(tmp_path):
    """Two ranks sharing one card: after one step the concatenated shards equal ONE process (the oracle) applying both
    ranks' gradients in a single sparse update per table -- device-side shard initialisation, HIP gathers / loss /
    one-launch updates, duplicates across ranks."""
    _run("_gpu_train_case", tmp_path)
    mf = importlib.import_module("matrix-factorization-torch_amd")
    mfd = mf.distributed
    world, std = 2, 1.0 / tdc.DIM**0.5
    ops = tdc.OracleOps()
    for opt in ("sgd", "adam"):
        ut = oembed.init_rows(tdc.N_USERS, tdc.DIM, 0, 1, 0, std)
        it = oembed.init_rows(tdc.N_ITEMS, tdc.DIM, 0, 1, 1, std)
        ut0, it0 = ut.clone(), it.clone()
        hyper = mfd.optimizer_hyper(opt, 0.05)
        u_ids, u_g, i_ids, i_g, losses = [], [], [], [], []
        for r in range(world):
            b = tdc._batch(r, world, mfd)
            loss, du, dv = ops.loss_and_grads("PairwiseLogisticLoss", oembed.gather(ut0, b["user"], True),
                                              oembed.gather(it0, b["item"], True), b["target"], b["item"], b["pos"], None, 0, 1.0, 1.0)
            u_ids.append(b["user"]); u_g.append(du); i_ids.append(b["item"]); i_g.append(dv); losses.append(loss)
        st = {"m": torch.zeros_like(ut), "v": torch.zeros_like(ut)}
        ops.update(opt, ut, st, torch.cat(u_ids), torch.cat(u_g), True, 1, hyper)
        st = {"m": torch.zeros_like(it), "v": torch.zeros_like(it)}
        ops.update(opt, it, st, torch.cat(i_ids), torch.cat(i_g), True, 1, hyper)
        got = [torch.load(f"{tmp_path}/{opt}_{r}.pt") for r in range(world)]
        # (the device's log / cos differ from libm's in the last bits of the initial rows: 1e-6; Adam's first step is
        # lr * sign-like: a sign flip of a ~0 gradient component would show as 2 lr, none occurs at this seed)
        torch.testing.assert_close(torch.cat([x["user"] for x in got]), ut, rtol=1e-4, atol=2e-5)
        items = torch.empty_like(it)
        for r in range(world):
            items[r::world] = got[r]["item"]
        torch.testing.assert_close(items, it, rtol=1e-4, atol=2e-5)
        for r in range(world):
            torch.testing.assert_close(got[r]["loss"], losses[r], rtol=1e-4, atol=1e-5)


def test_sharded_topk_world_2_with_hip_kernels(tmp_path):
    """Row-sharded catalog on two ranks, exclusion lists localised per shard, partial top-k merged: the bits of a full scan."""
    _run("_gpu_topk_case", tmp_path)
    g = torch.Generator().manual_seed(5)
    items = torch.nn.functional.normalize(torch.randn(tdc.N_ITEMS, tdc.DIM, generator=g), dim=-1)
    for r in range(2):
        got = torch.load(f"{tmp_path}/topk_{r}.pt")
        ws, wi = chain.topk(got["q"].numpy(), items.numpy(), tdc.K, got["excl"])
        assert np.array_equal(got["i"].numpy(), wi)
        assert np.array_equal(got["s"].numpy().view(np.uint32), ws.view(np.uint32))
