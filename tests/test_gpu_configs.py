"""The five BASELINE.json configurations as parity cases (MovieLens *shapes*, synthetic ids: the data
files do not exist offline), each through the module surface a user of the reference would call:
one training step (towers -> loss -> backward -> sparse update) against the CPU oracle, and the
retrieval each configuration names.  C3 is the bench workload (also covered at full size in
tests/test_gpu_module.py); C4 / C5 run their sharding / hashing at sizes the oracle finishes in
seconds."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import chain, embed as oembed, losses as ol

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _step_vs_oracle(mf, *, users, items, d, kind, b, p, optimizer, num_negatives=0, margin=1.0, logq=False, seed=0):
    """One step of module.training_step + optimizer.step on the GPU against the same step by the oracle."""
    cfg = {"num_users": users, "num_items": items, "hidden_size": d, "train_loss": kind, "num_negatives": num_negatives,
           "margin": margin, "optimizer": optimizer, "learning_rate": 0.05, "fused_losses": False, "use_logq": logq}
    torch.manual_seed(1234 + seed)                       # table initialisation
    m = mf.lightning.MatrixFactorizationLitModule(cfg)
    m.configure_model(device=DEV)
    data = mf.data.SyntheticInteractions(users, items, max_positives=p, seed=seed)
    batch = data.batch(b)
    if logq:
        m.logq = oembed.logq_from_counts(data.item_probability() * 1e6 + 1).to(DEV)
    ut, it = m.towers["user"].weight.detach().cpu().clone(), m.towers["item"].weight.detach().cpu().clone()
    opt = m.configure_optimizers()
    loss = m.training_step(mf.data.to_device(batch, DEV), 0)
    loss.backward()
    # what the backward parked on the tables for the sparse update: (row ids, gradient rows w.r.t. the unit rows)
    parked = {name: [(i.cpu(), g.cpu()) for i, g, _ in m.towers[name].weight._mf_pending] for name in ("user", "item")}
    opt.step()

    item_idx = torch.cat([batch["item"]["idx"], batch["neg_item"]["idx"]])
    u_raw, v_raw = ut[batch["user"]["idx"]].requires_grad_(), it[item_idx].requires_grad_()
    un, vn = oembed.gather(u_raw, torch.arange(b), True), oembed.gather(v_raw, torch.arange(2 * b), True)
    un.retain_grad()
    vn.retain_grad()
    lq = m.logq.cpu()[item_idx] if logq else None
    want = ol.loss(kind, un, vn, batch["target"], item_idx=item_idx, pos_idx=batch["user"]["pos_idx"],
                   num_negatives=num_negatives, sigma=1.0, margin=margin, logq=lq)
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-4 * max(1.0, abs(float(want))), (float(loss), float(want))
    # the summed gradient of every touched row -- what feeds the update, and for Adam's first step g / (|g| + eps)
    # the quantity whose last bits decide -- is bounded DIRECTLY, not through the updated tables
    for name, ids, g_or in (("user", batch["user"]["idx"], un.grad), ("item", item_idx, vn.grad)):
        (pid, pg), = parked[name]
        assert torch.equal(pid, ids)
        torch.testing.assert_close(pg, g_or, rtol=2e-4, atol=2e-6)
        rows_g, sum_g = oembed.coalesce(pid, pg)
        rows_o, sum_o = oembed.coalesce(ids, g_or)
        assert torch.equal(rows_g, rows_o)
        torch.testing.assert_close(sum_g, sum_o, rtol=2e-4, atol=1e-5)
    for table, ids, g in ((ut, batch["user"]["idx"], u_raw.grad), (it, item_idx, v_raw.grad)):
        if optimizer == "sgd":
            oembed.sgd_update(table, ids, g, 0.05, 0.0)
        else:
            oembed.adam_update(table, torch.zeros_like(table), torch.zeros_like(table), ids, g, step=1, lr=0.05,
                               weight_decay=0.01)
    _tables_close(m.towers["user"].weight.detach().cpu(), ut, batch["user"]["idx"], u_raw.grad, adam=optimizer == "adam")
    _tables_close(m.towers["item"].weight.detach().cpu(), it, item_idx, v_raw.grad, adam=optimizer == "adam")
    return m


def _tables_close(got, want, ids, grad, *, adam, lr=0.05):
    """SGD: elementwise.  Adam's first step moves a coordinate by lr * g / (|g| + eps), eps = 1e-8: where the
    summed gradient g is within fp32 summation noise of eps, its last bits (even its sign) decide the step,
    so those coordinates are only required to stay within the step bound; all others must agree elementwise."""
    if not adam:
        torch.testing.assert_close(got, want, rtol=1e-4, atol=2e-6)
        return
    uniq, acc = oembed.coalesce(ids, grad)
    ill = torch.zeros_like(got, dtype=torch.bool)
    ill[uniq] = acc.abs() < 1e-5                       # |g| >= 1e-5: g / (|g| + 1e-8) is 1 to 1e-3 whatever the noise
    assert torch.allclose(got[~ill], want[~ill], rtol=1e-4, atol=2e-6 + 2e-3 * lr)
    assert float(ill.float().mean()) < 0.02
    assert float((got - want).abs().max()) <= 2.0 * lr + 1e-6


def test_c1_ml100k_d32_bpr(mf):
    """configs[0]: MovieLens-100K shape, d = 32, BPR = PairwiseLogisticLoss(margin = 0)."""
    _step_vs_oracle(mf, users=944, items=1683, d=32, kind="PairwiseLogisticLoss", margin=0.0, b=512, p=20, optimizer="sgd")


def test_c2_ml1m_d64_sampled_softmax_and_full_catalog_topk(mf):
    """configs[1]: MovieLens-1M shape, d = 64, in-batch sampled softmax (InfoNCE), then exact top-20 over
    the whole 3,883-item catalog with the user's history excluded -- bit-exact indices and scores."""
    m = _step_vs_oracle(mf, users=6041, items=3884, d=64, kind="InfomationNoiseContrastiveEstimationLoss", b=1024, p=32,
                        optimizer="adam")
    m.on_validation_start()
    g = torch.Generator().manual_seed(1)
    users = torch.randint(1, 6041, (64,), generator=g)
    hist = [sorted(set(torch.randint(1, 3884, (int(n),), generator=g).tolist())) for n in torch.randint(0, 200, (64,), generator=g)]
    q = m(users.to(DEV), tower="user")
    s, i = m.item_processor.index.search(q, 20, exclude=hist)
    items = m.item_processor.index.embeddings.cpu().numpy()
    ws, wi = chain.topk(q.detach().cpu().numpy(), items, 20, hist)
    assert np.array_equal(i.cpu().numpy(), wi)
    assert np.array_equal(s.cpu().numpy().view(np.uint32), ws.view(np.uint32))


def test_c3_ml25m_d128_infonce_logq(mf):
    """configs[2] (the bench workload) at a batch the oracle handles: InfoNCE + logQ, row-Adam, d = 128."""
    _step_vs_oracle(mf, users=20_000, items=62_424, d=128, kind="InfomationNoiseContrastiveEstimationLoss", b=768, p=64,
                    optimizer="adam", logq=True)


def test_c4_row_sharded_catalog_topk_merge_is_bit_identical(mf):
    """configs[3]: the catalog dealt round-robin over 8 shards, per-shard top-k, merge of the partial
    results (what ShardedIndex does around its collectives) == one scan of the whole catalog."""
    g = torch.Generator().manual_seed(4)
    n, d, k, world = 62_423, 128, 20, 8
    items = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1).to(DEV)
    q = torch.nn.functional.normalize(torch.randn(96, d, generator=g), dim=-1).to(DEV)
    excl = [sorted(set(torch.randint(0, n, (int(c),), generator=g).tolist())) for c in torch.randint(0, 300, (96,), generator=g)]
    ref_s, ref_i = mf.retrieval.ItemIndex(items).search(q, k, exclude=excl)
    ps, pi = [], []
    for r in range(world):
        local_excl = [[(e - r) // world for e in ex if e % world == r] for ex in excl]
        s, i = mf.retrieval.ItemIndex(items[r::world].contiguous()).search(q, k, exclude=local_excl)
        ps.append(s)
        pi.append(torch.where(i >= 0, i * world + r, i))
    s, i = mf.retrieval.merge_topk(torch.stack(ps), torch.stack(pi), k)
    assert torch.equal(i, ref_i) and torch.equal(s, ref_s)


def test_c5_hash_bloom_towers_d256_training_step(mf):
    """configs[4] at reduced size: ids far beyond the table height, 2 hashes per id, d = 256; loss and the
    updated bucket rows against the oracle."""
    g = torch.Generator().manual_seed(5)
    buckets_u, buckets_i, d, b, nh = 5_000, 20_000, 256, 256, 2
    cfg = mf.models.ModelConfig(num_users=buckets_u, num_items=buckets_i, hidden_size=d, num_hashes=nh, hash_seed=3)
    towers = mf.models.init_towers(cfg, device=DEV)
    opt = mf.optim.SparseSGD(towers.parameters(), lr=0.05)
    user = torch.randint(0, 10_000_000, (b,), generator=g)
    item = torch.randint(0, 100_000_000, (2 * b,), generator=g)
    item[b: b + 10] = item[:10]                                   # some negatives collide with positives
    target = torch.randint(1, 6, (b,), generator=g)
    pos = torch.randint(0, 100_000_000, (b, 8), generator=g)
    pos[:, 0] = item[:b]
    tu, ti = towers["user"].weight.detach().cpu().clone(), towers["item"].weight.detach().cpu().clone()
    fn = mf.losses.InfomationNoiseContrastiveEstimationLoss()
    loss = fn(towers["user"](user.to(DEV)), towers["item"](item.to(DEV)), target.to(DEV), item_idx=item.to(DEV),
              pos_idx=pos.to(DEV))
    loss.backward()
    opt.step()
    bu, bi = oembed.hash_buckets(user, nh, 3, buckets_u), oembed.hash_buckets(item, nh, 4, buckets_i)
    raw_u = (tu[bu[:, 0]] + tu[bu[:, 1]]).requires_grad_()
    raw_i = (ti[bi[:, 0]] + ti[bi[:, 1]]).requires_grad_()
    un = raw_u / raw_u.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    vn = raw_i / raw_i.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    want = ol.loss("InfomationNoiseContrastiveEstimationLoss", un, vn, target, item_idx=item, pos_idx=pos)
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-4 * max(1.0, abs(float(want)))
    oembed.sgd_update(tu, bu.reshape(-1), raw_u.grad.repeat_interleave(nh, dim=0), 0.05, 0.0)
    oembed.sgd_update(ti, bi.reshape(-1), raw_i.grad.repeat_interleave(nh, dim=0), 0.05, 0.0)
    torch.testing.assert_close(towers["user"].weight.detach().cpu(), tu, rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(towers["item"].weight.detach().cpu(), ti, rtol=1e-4, atol=2e-6)
