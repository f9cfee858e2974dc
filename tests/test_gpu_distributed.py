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

from oracle import chain
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
        for tag in ("exact", "padded"):
            tr = mfd.ShardedTrainer(mf, DEV, opt, 0, num_users=tdc.N_USERS, num_items=tdc.N_ITEMS, dim=tdc.DIM, lr=0.05,
                                    kind="PairwiseLogisticLoss", comm=_relay_comm(mfd))        # user_mode "routed": any user anywhere
            assert isinstance(tr.ops, mfd.HipOps)
            b = {k: v.to(DEV) for k, v in tdc._batch(rank, world, mfd).items()}
            if tag == "exact":
                tr.prefetch(b)                       # exact counts (host read on the plan stream); else capacity-padded
            loss = tr.step(b, next_b=b)
            assert tr.padded_steps == (0 if tag == "exact" else 1)
            tr.finish()
            torch.cuda.synchronize()
            torch.save({"user": tr.user_table.cpu(), "item": tr.item_table.cpu(), "loss": loss.cpu()}, f"{out_dir}/{opt}_{tag}_{rank}.pt")


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
    """Two ranks sharing one card, EXAMPLE-sharded batches (any user on any rank): after one step the shards equal ONE
    process (the oracle) applying both ranks' gradients in a single sparse update per table -- device-side shard
    initialisation, the fused user + item exchange (exact and capacity-padded), HIP gathers (-1 requests: zero rows) /
    loss / one-launch updates (-1 ids skipped), duplicates across ranks.
    (The device's log / cos differ from libm's in the last bits of the initial rows: 1e-6; Adam's first step is
    lr * sign-like: a sign flip of a ~0 gradient component would show as 2 lr, none occurs at this seed.)"""
    _run("_gpu_train_case", tmp_path)
    tdc._check_train(tmp_path, 2, tags=("exact", "padded"), rtol=1e-4, atol=2e-5)


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
