"""GPU: the Lightning-module surface end to end, and size-independent properties at the
BASELINE configurations' full sizes (where the scalar oracle would take too long)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import embed as oembed, losses as ol
from tests import _golden_util as gu
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _module(mf, **over):
    cfg = {"num_users": 500, "num_items": 800, "hidden_size": 64, "learning_rate": 0.05, **over}
    m = mf.lightning.MatrixFactorizationLitModule(cfg)
    m.configure_model(device=DEV)
    return m


def test_compute_losses_keys_and_values_match_oracle(mf):
    m = _module(mf, num_negatives=4)
    batch = mf.data.to_device(mf.data.SyntheticInteractions(500, 800, max_positives=12, seed=3).batch(96), DEV)
    out = m.compute_losses(batch, step_name="train")
    assert list(out) == [f"train/{k}" for k in ol.KINDS]                 # xfmr_rec/lightning.py:137-146
    item_idx = torch.cat([batch["item"]["idx"], batch["neg_item"]["idx"]])
    u = m(batch["user"]["idx"], tower="user").detach().cpu()
    v = m(item_idx, tower="item").detach().cpu()
    want = ol.all_losses(u, v, batch["target"].cpu(), item_idx=item_idx.cpu(), pos_idx=batch["user"]["pos_idx"].cpu(),
                         num_negatives=4)
    for k in ol.KINDS:
        assert abs(float(out[f"train/{k}"]) - float(want[k])) <= 1e-4 * max(1.0, abs(float(want[k]))), k
    m.config.fused_losses = False
    single = m.compute_losses(batch)
    for k in ol.KINDS:
        assert float(single[f"train/{k}"]) == float(out[f"train/{k}"]), k


@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_training_loop_reduces_loss_and_recommend_excludes_history(mf, opt):
    m = _module(mf, optimizer=opt, train_loss="PairwiseLogisticLoss", margin=0.0, num_negatives=0)   # BPR (config C1)
    optim = m.configure_optimizers()
    data = mf.data.SyntheticInteractions(500, 800, max_positives=8, seed=1)
    fixed = mf.data.to_device(data.batch(256), DEV)
    first = last = None
    for step in range(30):
        loss = m.training_step(fixed, step)
        loss.backward()
        optim.step()
        optim.zero_grad()
        first = float(loss) if first is None else first
        last = float(loss)
    assert last < 0.9 * first, (first, last)
    m.on_validation_start()
    m.history[7] = [3, 4, 5]
    df = m.recommend(7, top_k=10, exclude_item_ids=[6])
    assert len(df) == 10 and not set(df["movie_id"]) & {3, 4, 5, 6}
    assert df["score"].is_monotonic_decreasing and float(df["score"].max()) <= 1.0 + 1e-5


def test_save_load_roundtrip_and_serving_shims(mf, tmp_path):
    """save() -> load(): same tables, same index, same recommendations; item-to-item excludes the item."""
    m = _module(mf)
    m.on_validation_start()
    m.history[9] = [11, 12]
    before = m.recommend_with_user_id(9, top_k=15, exclude_item_ids=[3])
    m.save(tmp_path / "model")
    assert {p.name for p in (tmp_path / "model").iterdir()} == {"towers.safetensors", "processors.json", "item_index.safetensors"}
    m2 = mf.lightning.MatrixFactorizationLitModule.load(tmp_path / "model", device=DEV)
    assert torch.equal(m2.towers["item"].weight, m.towers["item"].weight) and m2.history == m.history
    after = m2.recommend_with_user_id(9, top_k=15, exclude_item_ids=[3])
    assert before.equals(after) and not set(after["movie_id"]) & {3, 11, 12}
    sim = m2.recommend_with_item_id(42, top_k=10)
    assert len(sim) == 10 and 42 not in set(sim["movie_id"]) and sim["score"].is_monotonic_decreasing
    with pytest.raises(KeyError):
        m2.recommend_with_item_id(10**9)


def test_validation_step_metrics_match_oracle(mf):
    """validation_step = batched top-k (history excluded) + the six retrieval metrics, all on the device."""
    from oracle import retrieval as oretr

    m = _module(mf, top_k=10)
    m.on_validation_start()
    g = torch.Generator().manual_seed(4)
    users = torch.arange(1, 41)
    n_items = m.towers["item"].num_embeddings
    hist = [torch.randperm(n_items, generator=g)[: int(torch.randint(0, 30, (1,), generator=g))].sort().values for _ in users]
    tgts = [dict(zip(torch.randperm(n_items, generator=g)[:12].tolist(), torch.randint(1, 6, (12,), generator=g).float().tolist()))
            for _ in users]
    with torch.no_grad():                               # plant hits: a user's vector points at some of its targets
        iw = m.towers["item"].weight
        for r, t in enumerate(tgts):
            m.towers["user"].weight[users[r]] = iw[list(t)[:3]].sum(dim=0)
    csr = lambda lists: (torch.tensor([0] + list(np.cumsum([len(x) for x in lists])), device=DEV),  # noqa: E731
                         torch.cat([torch.as_tensor(list(x), dtype=torch.int64) for x in lists]).to(DEV))
    t_off, t_ids = csr([list(t) for t in tgts])
    t_rel = torch.tensor([v for t in tgts for v in t.values()], device=DEV)
    batch = {"user": {"idx": users.to(DEV)}, "history": csr(hist), "target": (t_off, t_ids, t_rel)}
    m.validation_step(batch, 0)
    got = m.metrics["val"].compute()
    _, rows = m.predict_step(batch, 0)
    assert not any(set(rows[r].tolist()) & set(hist[r].tolist()) for r in range(len(users)))
    want = oretr.retrieval_metrics(rows.cpu().numpy(), tgts, 10).mean(axis=0)
    assert list(got) == ["val/" + n for n in oretr.METRIC_NAMES]
    np.testing.assert_allclose([float(v) for v in got.values()], want, rtol=1e-5, atol=1e-6)
    assert float(got["val/RetrievalHitRate"]) > 0.5      # the planted targets are found


def _torch_infonce(u, v, target, item_idx, pos_idx, logq):
    """Plain torch fp32 reference on the GPU for full-size checks (dense, no mining)."""
    b = u.shape[0]
    lg = -0.5 * (torch.cdist(u, v) ** 2) * torch.sign(target.to(u.dtype))[:, None] - logq.to(u.dtype)[None, :]
    hit = item_idx[:b, None] == item_idx[None, :]
    for p in range(pos_idx.shape[1]):
        hit |= pos_idx[:, p, None] == item_idx[None, :]
    keep = ~hit
    keep[torch.arange(b), torch.arange(b)] = True
    lse = torch.logsumexp(torch.where(keep, lg, torch.full_like(lg, -float("inf"))), dim=1)
    return ((lse - lg.diagonal()) * target.abs()).sum()


def test_full_size_training_step_properties(mf):
    """Config C3 shapes (B = 8192, N = 16384, d = 128, P = 64): loss and gradients against a torch
    fp32 reference on the GPU, linearity of the backward in grad_out, finite outputs."""
    g = torch.Generator().manual_seed(0)
    b, n, d, p = 8192, 16384, 128, 64
    u0 = torch.nn.functional.normalize(torch.randn(b, d, generator=g), dim=-1).to(DEV)
    v0 = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1).to(DEV)
    target = torch.randint(1, 6, (b,), generator=g).to(DEV)
    w = 1.0 / torch.arange(1, 62423, dtype=torch.float64)
    item_idx = torch.cat([torch.multinomial(w, b, replacement=True, generator=g) + 1,
                          torch.randint(1, 62423, (b,), generator=g)]).to(DEV)
    pos_idx = (torch.multinomial(w, b * p, replacement=True, generator=g).reshape(b, p) + 1).to(DEV)
    pos_idx[:, 0] = item_idx[:b]
    logq = torch.log(torch.rand(n, generator=g) * 0.5 + 0.01).to(DEV)
    fn = mf.losses.InfomationNoiseContrastiveEstimationLoss(num_negatives=0)
    u, v = u0.clone().requires_grad_(), v0.clone().requires_grad_()
    loss = fn(u, v, target, item_idx=item_idx, pos_idx=pos_idx, logq=logq)
    loss.backward()
    ur, vr = u0.clone().requires_grad_(), v0.clone().requires_grad_()
    ref = _torch_infonce(ur, vr, target, item_idx, pos_idx, logq)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-4 * abs(float(ref)), (float(loss), float(ref))
    torch.testing.assert_close(u.grad, ur.grad, rtol=2e-3, atol=2e-5)
    torch.testing.assert_close(v.grad, vr.grad, rtol=2e-3, atol=2e-5)
    # ... and at the oracle's own tolerance (rtol 2e-4) against the same formulas in fp64 -- the fp32 torch restatement above
    # carries cdist's own rounding; every 128th user row and every 256th item row (64 + 64 rows)
    del ur, vr, ref
    ud, vd = u0.double().requires_grad_(), v0.double().requires_grad_()
    ref64 = _torch_infonce(ud, vd, target, item_idx, pos_idx, logq)
    ref64.backward()
    assert abs(float(loss) - float(ref64)) <= 2e-5 * abs(float(ref64)), (float(loss), float(ref64))
    # (row-wise: |err| <= 2e-4 (|x| + the row's largest entry) -- an entry is a sum of 16,384 fp32 terms of the row's scale)
    gu.assert_grads_close(u.grad[::128].cpu().numpy(), ud.grad[::128].cpu().numpy(), 1.0, "du vs fp64", base=2e-4, per_sigma=0.0, floor=0.0)
    gu.assert_grads_close(v.grad[::256].cpu().numpy(), vd.grad[::256].cpu().numpy(), 1.0, "dv vs fp64", base=2e-4, per_sigma=0.0, floor=0.0)
    del ud, vd, ref64
    u2, v2 = u0.clone().requires_grad_(), v0.clone().requires_grad_()
    (2.5 * fn(u2, v2, target, item_idx=item_idx, pos_idx=pos_idx, logq=logq)).backward()
    torch.testing.assert_close(u2.grad, 2.5 * u.grad, rtol=1e-3, atol=1e-4)      # backward is linear in grad_out
    assert torch.isfinite(u.grad).all() and torch.isfinite(v.grad).all()


def test_full_size_topk_properties(mf):
    """ML-25M catalog (62,423 x 128), Q = 1024, k = 20: sorted, excluded rows absent, scores agree
    with a torch matmul, idempotent, and the 8-shard merge reproduces the single scan bit for bit."""
    g = torch.Generator().manual_seed(1)
    n, d, q, k = 62423, 128, 1024, 20
    items = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1).to(DEV)
    queries = torch.nn.functional.normalize(torch.randn(q, d, generator=g), dim=-1).to(DEV)
    excl = [torch.unique(torch.randint(0, n, (int(m),), generator=g)).tolist()
            for m in torch.randint(0, 200, (q,), generator=g)]
    index = mf.retrieval.ItemIndex(items)
    s, i = index.search(queries, k, exclude=excl)
    s2, i2 = index.search(queries, k, exclude=excl)
    assert torch.equal(i, i2) and torch.equal(s, s2)
    assert (s[:, :-1] >= s[:, 1:]).all()
    ties = s[:, :-1] == s[:, 1:]
    assert (i[:, :-1][ties] < i[:, 1:][ties]).all()
    full = queries @ items.T
    for r, ex in enumerate(excl):
        if ex:
            full[r, torch.tensor(ex, device=DEV)] = -float("inf")
    ts, ti = torch.topk(full, k, dim=1)
    torch.testing.assert_close(s, ts, rtol=0, atol=2e-6)
    agree = (i == ti).float().mean()
    assert agree > 0.999, float(agree)                       # matmul order differs in the last bit near ties
    assert all(not set(i[r].tolist()) & set(excl[r]) for r in range(0, q, 37))
    bounds = np.linspace(0, n, 9).astype(int)
    ps, pi = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        a, b = mf.retrieval.ItemIndex(items[lo:hi], idx_base=int(lo)).search(queries, k, exclude=excl)
        ps.append(a)
        pi.append(b)
    ms, mi = mf.retrieval.merge_topk(torch.stack(ps), torch.stack(pi), k)
    assert torch.equal(mi, i) and torch.equal(ms, s)


def test_sharded_classes_on_one_rank_use_the_hip_path(mf):
    """world_size 1 over RCCL: ShardedTrainer / ShardedIndex with HipOps reproduce the single-GPU API."""
    import os
    import socket

    import torch.distributed as dist

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        nu, ni, d, b = 300, 500, 64, 64
        tr = mf.distributed.ShardedTrainer(mf, DEV, "sgd", 0, num_users=nu, num_items=ni, dim=d, lr=0.1,
                                           kind="InfomationNoiseContrastiveEstimationLoss")
        g = torch.Generator().manual_seed(2)
        batch = {"user": torch.randint(0, nu, (b,), generator=g).to(DEV), "item": torch.randint(0, ni, (2 * b,), generator=g).to(DEV),
                 "target": torch.randint(1, 6, (b,), generator=g).to(DEV), "pos": torch.randint(0, ni, (b, 5), generator=g).to(DEV)}
        towers = mf.models.init_towers(mf.models.ModelConfig(num_users=nu, num_items=ni, hidden_size=d), device=DEV)
        with torch.no_grad():
            towers["user"].weight.copy_(tr.user_table)
            towers["item"].weight.copy_(tr.item_table)
        opt = mf.optim.SparseSGD(towers.parameters(), lr=0.1)
        fn = mf.losses.InfomationNoiseContrastiveEstimationLoss(num_negatives=0)
        want = fn(towers["user"](batch["user"]), towers["item"](batch["item"]), batch["target"], item_idx=batch["item"],
                  pos_idx=batch["pos"])
        want.backward()
        opt.step()
        got = tr.step(batch, next_b=batch)       # no plan yet: capacity-padded exchanges (no host read); the next batch's
        assert float(got) == float(want)         # exact plan is prefetched on the side stream meanwhile
        assert tr.padded_steps == 1
        torch.testing.assert_close(tr.item_table, towers["item"].weight.detach(), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(tr.user_table, towers["user"].weight.detach(), rtol=1e-6, atol=1e-7)
        assert isinstance(tr.comm, mf.distributed.RcclComm)           # the exchanges ran through mf_comm_* on the compute stream
        assert tr.comm.rccl_ranks == 1 and tr.comm.transport == "mf_comm"
        # libmf_hip.so took the librccl torch had already mapped (RTLD_NOLOAD): one RCCL runtime per process
        assert mf._lib.lib().mf_comm_source().decode().startswith("shared"), mf._lib.lib().mf_comm_source()
        want2 = fn(towers["user"](batch["user"]), towers["item"](batch["item"]), batch["target"], item_idx=batch["item"],
                   pos_idx=batch["pos"])
        want2.backward()
        opt.step()
        got2 = tr.step(batch)                    # the prefetched exact plan
        tr.finish()
        assert float(got2) == float(want2) and tr.padded_steps == 1
        torch.testing.assert_close(tr.item_table, towers["item"].weight.detach(), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(tr.user_table, towers["user"].weight.detach(), rtol=1e-6, atol=1e-7)
        q = tr.user_vectors(torch.arange(1, 9, device=DEV))
        s1, i1 = mf.distributed.ShardedIndex(tr.item_shard(), tr.item_offset(), ni, stride=tr.item_stride()).search(q, 10)
        s2, i2 = mf.retrieval.ItemIndex(tr.item_shard()).search(q, 10)
        assert torch.equal(i1, i2) and torch.equal(s1, s2)
        # shards are generated on the device: the counter-based generator against its numpy restatement
        from oracle import embed as oembed

        for start, stride, n_loc in ((0, 1, 300), (3, 8, 41)):
            got = mf.distributed.HipOps(mf).init_rows(n_loc, d, start, stride, 5, 0.125, DEV)
            torch.testing.assert_close(got.cpu(), oembed.init_rows(n_loc, d, start, stride, 5, 0.125), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(tr.user_table.cpu() * 0 + mf.distributed.HipOps(mf).init_rows(nu, d, 0, 1, 0, 1 / d**0.5, DEV).cpu(),
                                   oembed.init_rows(nu, d, 0, 1, 0, 1 / d**0.5), rtol=1e-5, atol=1e-6)
        # hash / bloom towers through the sharded step (one rank: every bucket row is "remote" to itself) ==
        # the single-GPU HashEmbeddingTower step on the same bucket tables
        bu, bi, nh = 700, 1900, 2
        trh = mf.distributed.ShardedTrainer(mf, DEV, "sgd", 0, num_users=bu, num_items=bi, dim=d, lr=0.1, num_hashes=nh, hash_seed=3)
        towers = mf.models.init_towers(mf.models.ModelConfig(num_users=bu, num_items=bi, hidden_size=d, num_hashes=nh, hash_seed=3),
                                       device=DEV)
        with torch.no_grad():
            towers["user"].weight.copy_(trh.user_table)
            towers["item"].weight.copy_(trh.item_table)
        hb = {"user": torch.randint(0, 10_000_000, (b,), generator=g).to(DEV), "item": torch.randint(0, 100_000_000, (2 * b,), generator=g).to(DEV),
              "target": batch["target"], "pos": torch.randint(0, 100_000_000, (b, 5), generator=g).to(DEV)}
        opt = mf.optim.SparseSGD(towers.parameters(), lr=0.1)
        want = fn(towers["user"](hb["user"]), towers["item"](hb["item"]), hb["target"], item_idx=hb["item"], pos_idx=hb["pos"])
        want.backward()
        opt.step()
        got = trh.step(hb)
        torch.testing.assert_close(got, want.detach(), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(trh.item_table, towers["item"].weight.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(trh.user_table, towers["user"].weight.detach(), rtol=1e-5, atol=1e-6)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cfg", [("adam", "PairwiseHingeLoss", 4, 32), ("adam", "InfomationNoiseContrastiveEstimationLoss", 0, 256),
                                 ("sgd", "PairwiseLogisticLoss", 0, 64)], ids=lambda c: "-".join(map(str, c)))
def test_captured_step_replays_bit_identically(mf, cfg):
    """A whole training step (gathers -> loss -> backward -> sparse updates) recorded in a hipGraph and replayed on
    new batches == the same steps run eagerly, bit for bit: tables, moments and the returned loss.  First case: the
    reference's default configuration -- PairwiseHingeLoss, num_negatives = 4, BATCH_SIZE = 32
    (xfmr_rec/lightning.py:38-39, params.py:18)."""
    opt_name, kind, k, b = cfg
    users, items, d = 400, 900, 64

    def build():
        torch.manual_seed(7)
        towers = mf.models.init_towers(mf.models.ModelConfig(num_users=users, num_items=items, hidden_size=d), device=DEV)
        opt = (mf.optim.RowAdam(towers.parameters(), lr=0.01) if opt_name == "adam"
               else mf.optim.SparseSGD(towers.parameters(), lr=0.05))
        fn = getattr(mf.losses, kind)(num_negatives=k)
        one = torch.ones((), device=DEV)

        def step(bt):
            loss = fn(towers["user"](bt["user"]), towers["item"](bt["item"]), bt["target"], item_idx=bt["item"], pos_idx=bt["pos"])
            loss.backward(one)
            opt.step()
            return loss.detach()

        return towers, opt, step

    g = torch.Generator().manual_seed(b)
    batches = []
    for _ in range(4):
        item = torch.randint(1, items, (2 * b,), generator=g)
        pos = torch.randint(0, items, (b, 6), generator=g)
        pos[:, 0] = item[:b]
        batches.append({name: x.to(DEV) for name, x in dict(user=torch.randint(1, users, (b,), generator=g), item=item,
                                                              target=torch.randint(1, 6, (b,), generator=g), pos=pos).items()})
    # eager: three steps on batch 0 (what the capture's warm-up does), then batches 1..3
    towers_e, opt_e, step_e = build()
    losses_e = [step_e(bt) for bt in [batches[0]] * 3 + batches[1:]]
    # captured
    towers_g, opt_g, step_g = build()
    captured = mf.graph.CapturedStep(step_g, batches[0], optimizers=[opt_g], warmup=3)
    losses_g = [captured(bt).clone() for bt in batches[1:]]
    assert captured.replays == 3
    for name in ("user", "item"):
        assert torch.equal(towers_g[name].weight, towers_e[name].weight), name
    for le, lg in zip(losses_e[3:], losses_g):
        assert torch.equal(le, lg)
    if opt_name == "adam":
        for pe, pg in zip(opt_e.param_groups[0]["params"], opt_g.param_groups[0]["params"]):
            assert torch.equal(opt_e.state[pe]["exp_avg"], opt_g.state[pg]["exp_avg"])
            assert opt_e.state[pe]["step"] == opt_g.state[pg]["step"] == 6
    with pytest.raises(ValueError, match="ONE shape"):
        captured({**batches[1], "user": batches[1]["user"][:-1]})


@pytest.mark.parametrize("cfg", [("adam", "PairwiseHingeLoss", 4, 32, 32, False, False), ("adam", "PairwiseHingeLoss", 4, 32, 128, True, False),
                                 ("sgd", "InfomationNoiseContrastiveEstimationLoss", 8, 128, 64, False, True),
                                 ("adam", "MutualInformationNeuralEstimationLoss", 64, 100, 256, True, True),
                                 ("sgd", "ContrastiveLoss", 1, 5, 32, False, False),
                                 ("adam", "AlignmentContrastiveLoss", 3, 7, 32, False, True),           # N = 14: less than one tile
                                 ("sgd", "PairwiseLogisticLoss", 2, 64, 128, True, False),
                                 ("adam", "InfomationNoiseContrastiveEstimationLoss", 16, 128, 32, False, False),    # the largest batch
                                 # the tuners' far end (ray.py:147-149): sigma = 1000 moves the fixed-point dV unit off 2^-40 (dv_fix_of)
                                 ("sgd", "MutualInformationNeuralEstimationLoss", 4, 128, 32, False, False, 1000.0, -0.5),
                                 ("adam", "PairwiseLogisticLoss", 32, 96, 64, True, False, 1000.0, 1.0)],
                         ids=lambda c: "-".join(map(str, c)))
def test_fused_small_step_is_bit_identical_to_the_multi_kernel_step(mf, cfg):
    """The reference's default step (B = 32, PairwiseHinge, 4 mined negatives, AdamW; xfmr_rec/params.py:18, lightning.py:38-39)
    in ONE launch (mf_step_small) against the ordinary sequence gather -> loss -> backward -> update: the same loss value and --
    torch.equal -- the same tables and Adam moments after three steps, with duplicate users / items inside a batch, zero and
    negative targets, padded and CSR positives, a logQ table, k = 1 and k = 64, B up to 128."""
    opt_name, kind, k, b, d, use_csr, use_logq = cfg[:7]
    sigma, margin = cfg[7:] if len(cfg) > 7 else (1.3, 0.7)
    n_users, n_items = 60, 90
    g = torch.Generator().manual_seed(b * 7 + d)

    def make():
        towers = mf.models.init_towers(mf.models.ModelConfig(num_users=n_users, num_items=n_items, hidden_size=d), device=DEV)
        opt = mf.optim.RowAdam(towers.parameters(), lr=0.05) if opt_name == "adam" else mf.optim.SparseSGD(towers.parameters(), lr=0.1, weight_decay=0.01)
        return towers, opt

    ta, oa = make()
    tb, ob = make()
    with torch.no_grad():
        for name in ("user", "item"):
            tb[name].weight.copy_(ta[name].weight)
    fn = getattr(mf.losses, kind)(num_negatives=k, sigma=sigma, margin=margin)
    logq = (torch.rand(n_items, generator=g) - 0.5).to(DEV) if use_logq else None
    lists = [torch.randperm(n_items - 1, generator=g)[: int(ln)] + 1 for ln in torch.randint(0, 12, (n_users,), generator=g)]
    off = torch.tensor([0] + list(np.cumsum([x.numel() for x in lists])), dtype=torch.int64).to(DEV)
    flat = torch.cat(lists).to(DEV)
    fused = mf.fused.FusedSmallStep(tb, ob, fn, logq_table=logq)
    one = torch.ones((), device=DEV)
    for step in range(3):
        user = torch.randint(1, n_users, (b,), generator=g)
        item = torch.randint(1, n_items, (2 * b,), generator=g)
        if b >= 4:
            user[1] = user[0]
            item[b] = item[0]                     # a sampled negative that is row 0's positive
            item[2] = item[3]
        target = torch.randint(-1, 6, (b,), generator=g)
        batch = {"user": user.to(DEV), "item": item.to(DEV), "target": target.to(DEV)}
        if use_csr:
            batch["pos_csr"] = (batch["user"], off, flat)
        else:
            pos = torch.randint(0, n_items, (b, 6), generator=g)
            pos[:, 0] = item[:b]
            batch["pos"] = pos.to(DEV)
        # the ordinary path
        u = ta["user"](batch["user"])
        v = ta["item"](batch["item"])
        want = fn(u, v, batch["target"], item_idx=batch["item"], pos_idx=batch.get("pos"), logq_table=logq, pos_csr=batch.get("pos_csr"))
        want.backward(one)
        oa.step()
        got = fused(batch)
        assert fused.fused_steps == step + 1 and fused.fallback_steps == 0
        assert torch.equal(got, want.detach()) or (torch.isinf(got) and torch.isinf(want)), (step, float(got), float(want))
        for name in ("user", "item"):
            assert torch.equal(ta[name].weight, tb[name].weight), (step, name, float((ta[name].weight - tb[name].weight).abs().max()))
            if opt_name == "adam":
                sa, sb = oa.state[ta[name].weight], ob.state[tb[name].weight]
                assert sa["step"] == sb["step"]
                assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    # shapes outside the one-launch range fall back to the ordinary path, with the same results
    big = {"user": torch.randint(1, n_users, (200,), generator=g).to(DEV), "item": torch.randint(1, n_items, (400,), generator=g).to(DEV),
           "target": torch.ones(200, device=DEV), "pos": torch.zeros(200, 1, dtype=torch.int64, device=DEV)}
    assert not fused.supported(big)
    fused(big)
    assert fused.fallback_steps == 1


@pytest.mark.parametrize("path_kind", ["one_launch", "multi_kernel"])
@pytest.mark.parametrize("opt_name", ["sgd", "adam"])
def test_default_step_matches_the_reference_on_a_table_consistent_fixture(mf, opt_name, path_kind):
    """``mf_step_small`` (and the multi-kernel sequence it must equal) against the REFERENCE, not against another HIP path:
    tests/golden/step_B32_N64_d32_P16.npz holds raw tables, ids (duplicates share rows), and what xfmr_rec/losses.py +
    torch.autograd + torch.optim.SGD / AdamW make of them at the reference's default shape (params.py:18,
    lightning.py:33,38-39,189-192,238-239).  Per trained loss: all seven ``out_losses`` within 1e-4 of the reference's
    values; the tables after the step against the fixture's (SGD: 1e-5 abs; AdamW's first step is lr * g / (|g| + eps),
    compared where |g| > 1e-4), and against oracle.embed.sgd_update fed the reference's du / dv through
    oracle.embed.normalize_backward; untouched rows bit-unchanged."""
    z = np.load(GOLDEN / "step_B32_N64_d32_P16.npz")
    t = {k: torch.from_numpy(z[k]) for k in ("U", "V", "user", "item", "target", "pos_idx")}
    k = int(z["num_negatives"])
    batch = {"user": t["user"].to(DEV), "item": t["item"].to(DEV), "target": t["target"].to(DEV), "pos": t["pos_idx"].to(DEV)}
    one = torch.ones((), device=DEV)
    for ki, kind in enumerate(ol.KINDS):
        if ki == 0:
            continue                                  # AlignmentLoss has no negatives to mine: not a one-launch shape
        towers = mf.models.init_towers(mf.models.ModelConfig(num_users=t["U"].shape[0], num_items=t["V"].shape[0], hidden_size=32), device=DEV)
        with torch.no_grad():
            towers["user"].weight.copy_(t["U"].to(DEV))
            towers["item"].weight.copy_(t["V"].to(DEV))
        opt = (mf.optim.SparseSGD(towers.parameters(), lr=float(z["lr_sgd"])) if opt_name == "sgd"
               else mf.optim.RowAdam(towers.parameters(), lr=float(z["lr_adam"]), weight_decay=0.0))
        fn = getattr(mf.losses, kind)(num_negatives=k)
        if path_kind == "one_launch":
            step = mf.fused.FusedSmallStep(towers, opt, fn, all_losses=True)
            got = step(batch)
            assert step.fused_steps == 1 and step.fallback_steps == 0
            seven = step.losses.cpu().numpy()
            for kj, other in enumerate(ol.KINDS):
                want = float(z[f"loss_{kj}"])
                assert abs(float(seven[kj]) - want) <= 1e-4 * max(1.0, abs(want)), (kind, other, float(seven[kj]), want)
        else:
            got = fn(towers["user"](batch["user"]), towers["item"](batch["item"]), batch["target"], item_idx=batch["item"], pos_idx=batch["pos"])
            got.backward(one)
            opt.step()
        want = float(z[f"loss_{ki}"])
        assert abs(float(got) - want) <= 1e-4 * max(1.0, abs(want)), (kind, float(got), want)
        for name, ids, tab in (("U", t["user"], towers["user"].weight), ("V", t["item"], towers["item"].weight)):
            new = tab.detach().cpu()
            touched = torch.zeros(new.shape[0], dtype=torch.bool)
            touched[ids] = True
            assert torch.equal(new[~touched], t[name][~touched]), (kind, name)
            if opt_name == "sgd":
                np.testing.assert_allclose(new.numpy(), z[f"{name}_sgd_{ki}"], rtol=0, atol=1e-5, err_msg=f"{kind} {name}")
                small = "du" if name == "U" else "dv"
                raw = oembed.normalize_backward(t[name][ids], torch.from_numpy(z[f"{small}_{ki}"]))
                mine = t[name].clone()
                oembed.sgd_update(mine, ids, raw, float(z["lr_sgd"]))
                np.testing.assert_allclose(new.numpy(), mine.numpy(), rtol=0, atol=1e-5, err_msg=f"{kind} {name} (oracle update)")
            else:
                firm = np.abs(z[f"d{name}_{ki}"]) > 1e-4
                np.testing.assert_allclose(new.numpy()[firm], z[f"{name}_adam_{ki}"][firm], rtol=0, atol=2e-5, err_msg=f"{kind} {name}")
                assert firm.sum() > 0


@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_module_fused_training_step_equals_the_three_calls(mf, opt):
    """``fused_training_step`` (one launch) against ``training_step`` + ``backward`` + ``optimizer.step()`` at the reference's
    default configuration (B = 32, hidden_size = 32, PairwiseHinge, 4 mined negatives): the same seven logged losses and --
    torch.equal -- the same tables after three steps; a batch too large for one launch takes the three calls by itself."""
    seen = []

    def make():
        torch.manual_seed(11)
        m = mf.lightning.MatrixFactorizationLitModule({"num_users": 300, "num_items": 400, "learning_rate": 0.05, "optimizer": opt})
        m.configure_model(device=DEV)
        return m, m.configure_optimizers()

    (ma, oa), (mb, ob) = make(), make()
    assert ma.config.hidden_size == 32 and ma.config.train_loss == "PairwiseHingeLoss" and ma.config.num_negatives == 4
    mb.log_dict = lambda d, *a, **k: seen.append({key: float(v) for key, v in d.items()})
    data = mf.data.SyntheticInteractions(300, 400, max_positives=9, seed=5)
    for step, bsz in enumerate((32, 32, 32, 200)):
        batch = mf.data.to_device(data.batch(bsz), DEV)
        want = ma.compute_losses(batch, step_name="train")
        loss = want["train/PairwiseHingeLoss"]
        loss.backward()
        oa.step()
        oa.zero_grad(set_to_none=True)
        got = mb.fused_training_step(batch, ob)
        assert float(got) == float(loss), step
        for name in ("user", "item"):
            assert torch.equal(ma.towers[name].weight, mb.towers[name].weight), (step, name)
        if bsz <= 128:
            assert list(seen[-1]) == [f"train/{k}" for k in ol.KINDS]
            for k in ol.KINDS:
                assert seen[-1][f"train/{k}"] == float(want[f"train/{k}"]), (step, k)
    assert mb._fused.fused_steps == 3 and mb._fused.fallback_steps == 1


def test_fused_training_step_uses_the_logq_table_of_the_step(mf):
    """``self.logq`` set (None -> table) and refreshed AFTER the first fused step must reach the one-launch path like it
    reaches ``training_step``: tables torch.equal to the three calls at every step."""
    def make():
        torch.manual_seed(7)
        m = mf.lightning.MatrixFactorizationLitModule({"num_users": 200, "num_items": 300, "learning_rate": 0.05, "use_logq": True,
                                                       "train_loss": "InfomationNoiseContrastiveEstimationLoss"})
        m.configure_model(device=DEV)
        return m, m.configure_optimizers()

    (ma, oa), (mb, ob) = make(), make()
    data = mf.data.SyntheticInteractions(200, 300, max_positives=7, seed=9)
    g = torch.Generator().manual_seed(1)
    tables = [None, torch.rand(300, generator=g).log().to(DEV), torch.rand(300, generator=g).log().to(DEV)]
    for step, table in enumerate(tables):
        ma.logq = mb.logq = table
        batch = mf.data.to_device(data.batch(32), DEV)
        loss = ma.training_step(batch)
        loss.backward()
        oa.step()
        oa.zero_grad(set_to_none=True)
        got = mb.fused_training_step(batch, ob)
        assert float(got) == float(loss), step
        for name in ("user", "item"):
            assert torch.equal(ma.towers[name].weight, mb.towers[name].weight), (step, name)
    assert mb._fused.fused_steps == 3


@pytest.mark.parametrize("mode", ["csr", "padded"])
def test_fused_small_step_with_long_positive_lists(mf, mode):
    """Positive lists beyond the 512 entries a half-wave walks (a heavy user: thousands of positives; padded: P = 700) go
    to the whole workgroup of the one-launch step: tables torch.equal to the multi-kernel step, masks therefore too."""
    n_users, n_items, b, d = 40, 5000, 24, 32
    g = torch.Generator().manual_seed(3)

    def make():
        torch.manual_seed(5)
        towers = mf.models.init_towers(mf.models.ModelConfig(num_users=n_users, num_items=n_items, hidden_size=d), device=DEV)
        return towers, mf.optim.RowAdam(towers.parameters(), lr=0.05)

    (ta, oa), (tb, ob) = make(), make()
    fn = mf.losses.PairwiseHingeLoss(num_negatives=4)
    lens = torch.randint(0, 30, (n_users,), generator=g)
    lens[3], lens[7], lens[11] = 3000, 600, 513
    lists = [torch.randint(1, n_items, (int(ln),), generator=g) for ln in lens]
    off = torch.tensor([0] + torch.tensor([x.numel() for x in lists]).cumsum(0).tolist(), dtype=torch.int64).to(DEV)
    flat = torch.cat(lists).to(DEV)
    fused = mf.fused.FusedSmallStep(tb, ob, fn)
    one = torch.ones((), device=DEV)
    for step in range(2):
        user = torch.randint(1, n_users, (b,), generator=g)
        user[:3] = torch.tensor([3, 7, 11])
        item = torch.randint(1, n_items, (2 * b,), generator=g)
        item[b:b + 8] = lists[3][2000:2008]                      # negatives that are positives of the heavy user, deep in its list
        item[b + 8] = lists[7][599]
        batch = {"user": user.to(DEV), "item": item.to(DEV), "target": torch.randint(1, 6, (b,), generator=g).to(DEV)}
        if mode == "csr":
            batch["pos_csr"] = (batch["user"], off, flat)
        else:
            pos = torch.randint(1, n_items, (b, 700), generator=g)
            pos[0, 650:658] = item[b:b + 8]                      # beyond the half-wave's share
            pos[:, 0] = item[:b]
            batch["pos"] = pos.to(DEV)
        want = fn(ta["user"](batch["user"]), ta["item"](batch["item"]), batch["target"], item_idx=batch["item"], pos_idx=batch.get("pos"),
                  pos_csr=batch.get("pos_csr"))
        want.backward(one)
        oa.step()
        oa.zero_grad(set_to_none=True)
        got = fused(batch)
        assert fused.fallback_steps == 0
        assert torch.equal(got, want.detach()), step
        for name in ("user", "item"):
            assert torch.equal(ta[name].weight, tb[name].weight), (step, name)
