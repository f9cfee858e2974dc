"""CPU: pin the oracle modules the reference cannot pin (it has no embedding tables, no SGD / row Adam, no exact
top-k and torchmetrics is not installed here) against third-party definitions they do not themselves state:
torch.optim.SGD / AdamW, torch.nn.functional.embedding / normalize, an fp64 matmul + stable sort, hand-worked
values of the torchmetrics retrieval formulas, and an independent (tuple-sort) restatement of the two 64-bit key
orders that oracle/chain.c shares with the kernels through include/mf_numerics.h."""
from __future__ import annotations

import math

import numpy as np
import pytest
import torch

from oracle import chain, embed as oembed, retrieval as oretr


# ------------------------------------------------------------------ towers ---
@pytest.mark.parametrize("normalize", [False, True])
def test_gather_is_embedding_plus_normalize(normalize):
    g = torch.Generator().manual_seed(0)
    table = torch.randn(50, 24, generator=g)
    table[7] = 0.0                                           # zero row: the 1e-12 clamp decides
    idx = torch.randint(0, 50, (6, 5), generator=g)
    idx[0, 0] = 7
    want = torch.nn.functional.embedding(idx, table)
    if normalize:
        want = torch.nn.functional.normalize(want, p=2.0, dim=-1, eps=1e-12)
    torch.testing.assert_close(oembed.gather(table, idx, normalize), want, rtol=1e-6, atol=1e-7)


def _dense_grad(rows, d, idx, grad):
    out = torch.zeros(rows, d)
    out.index_add_(0, idx, grad)
    return out


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_sgd_update_is_torch_sgd_on_the_touched_rows(wd):
    """Three steps, duplicate ids in every batch.  torch.optim.SGD decays every row; the sparse update only the
    touched ones -- so the check is on rows touched in every step (all rows when wd = 0)."""
    g = torch.Generator().manual_seed(1)
    rows, d, lr = 40, 16, 0.1
    table0 = torch.randn(rows, d, generator=g)
    dense = torch.nn.Parameter(table0.clone())
    opt = torch.optim.SGD([dense], lr=lr, weight_decay=wd)
    mine = table0.clone()
    always = torch.arange(0, 12)
    for _ in range(3):
        idx = torch.cat([always, always[torch.randint(0, 12, (30,), generator=g)]])          # duplicates
        if wd == 0.0:
            idx = torch.cat([idx, torch.randint(12, rows, (9,), generator=g)])             # and rows touched now and then
        grad = torch.randn(idx.numel(), d, generator=g)
        dense.grad = _dense_grad(rows, d, idx, grad)
        opt.step()
        oembed.sgd_update(mine, idx, grad, lr, wd)
    check = torch.arange(rows) if wd == 0.0 else always
    torch.testing.assert_close(mine[check], dense.detach()[check], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_update_is_torch_adamw_on_rows_touched_every_step(wd):
    """The row-wise lazy AdamW equals dense torch.optim.AdamW (the reference's optimiser class,
    xfmr_rec/lightning.py:238-239) on every row that receives gradient in every step: same moments, same global
    bias correction, same decoupled decay.  (Rows that skip a step keep their moments here and decay there: the
    documented difference of a lazy update.)"""
    g = torch.Generator().manual_seed(2)
    rows, d, lr = 30, 8, 0.05
    table0 = torch.randn(rows, d, generator=g)
    dense = torch.nn.Parameter(table0.clone())
    opt = torch.optim.AdamW([dense], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    mine, m, v = table0.clone(), torch.zeros(rows, d), torch.zeros(rows, d)
    always = torch.arange(3, 17)
    for step in (1, 2, 3):
        idx = torch.cat([always, always[torch.randint(0, always.numel(), (25,), generator=g)]])
        grad = torch.randn(idx.numel(), d, generator=g)
        dense.grad = _dense_grad(rows, d, idx, grad)
        opt.step()
        oembed.adam_update(mine, m, v, idx, grad, step=step, lr=lr, weight_decay=wd)
    torch.testing.assert_close(mine[always], dense.detach()[always], rtol=2e-6, atol=2e-6)
    state = opt.state[dense]
    torch.testing.assert_close(m[always], state["exp_avg"][always], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(v[always], state["exp_avg_sq"][always], rtol=1e-6, atol=1e-7)
    untouched = torch.tensor([0, 1, 2, 20, 29])
    assert torch.equal(mine[untouched], table0[untouched])                                  # lazy: they do not move


# --------------------------------------------------------------- retrieval ---
def test_topk_exact_is_fp64_matmul_plus_stable_sort():
    """(score desc, row asc) of exact scores: on data whose top scores are separated by far more than fp32
    rounding the chain oracle must return the rows of an fp64 matmul + stable sort, exclusions removed."""
    g = torch.Generator().manual_seed(3)
    q = torch.nn.functional.normalize(torch.randn(9, 32, generator=g), dim=-1)
    items = torch.nn.functional.normalize(torch.randn(400, 32, generator=g), dim=-1)
    excl = [sorted(set(torch.randint(0, 400, (int(n),), generator=g).tolist())) for n in torch.randint(0, 30, (9,), generator=g)]
    k = 12
    s64 = q.double() @ items.double().T
    for r, ex in enumerate(excl):
        s64[r, ex] = -math.inf
    order = torch.argsort(s64, dim=1, descending=True, stable=True)[:, : k + 1]
    top = torch.gather(s64, 1, order)
    assert float((top[:, :-1] - top[:, 1:]).min()) > 1e-5            # tie-free: the order is unambiguous in fp32 too
    got_s, got_i = oretr.topk_exact(q.numpy(), items.numpy(), k, excl)
    assert np.array_equal(got_i, order[:, :k].numpy())
    np.testing.assert_allclose(got_s, top[:, :k].numpy(), rtol=0, atol=1e-6)


def test_topk_ties_go_to_the_lowest_row():
    items = np.zeros((10, 8), np.float32)
    items[[2, 5, 7], 0] = 1.0
    q = np.zeros((1, 8), np.float32)
    q[0, 0] = 1.0
    s, i = oretr.topk_exact(q, items, 5)
    assert i[0].tolist() == [2, 5, 7, 0, 1] and s[0].tolist() == [1.0, 1.0, 1.0, 0.0, 0.0]


def test_retrieval_metrics_hand_worked_torchmetrics_formulas():
    """Values worked out by hand from the torchmetrics (1.8) functional definitions with top_k = k:
    DCG = sum rel_i / log2(i + 1) (linear gain), NDCG = DCG@k / IDCG@k; recall = hits@k / #relevant;
    precision = hits@k / k; AP = mean over the relevant ranks r <= k of (#relevant up to r) / r; hit rate;
    reciprocal rank of the first relevant item.  Targets the search missed rank right below the retrieved items
    (xfmr_rec/lightning.py:170-175), relevance = rating, relevant = rating > 0."""
    l2 = math.log2
    cases = [
        # k = 3, retrieved [10, 20, 30], targets {20: 3, 40: 2}: rel@3 = [0, 3, 0]
        (3, [10, 20, 30], {20: 3.0, 40: 2.0},
         [(3 / l2(3)) / (3 / l2(2) + 2 / l2(3)), 1 / 2, 1 / 3, 1 / 2, 1.0, 1 / 2]),
        # fewer than k retrieved: [5, -1, -1]; the missed target 7 ranks second: rel@3 = [2, 1]
        (3, [5, -1, -1], {7: 1.0, 5: 2.0}, [1.0, 1.0, 2 / 3, 1.0, 1.0, 1.0]),
        # a rating-0 target is a target that is not relevant: rel@3 = [0, 0, 4]
        (3, [9, 1, 3], {9: 0.0, 3: 4.0}, [(4 / l2(4)) / (4 / l2(2)), 1.0, 1 / 3, 1 / 3, 1.0, 1 / 3]),
        # relevant item retrieved beyond k does not count: k = 2, retrieved [1, 2, 3] -> only [1, 2] looked at
        (2, [1, 2], {3: 5.0, 8: 1.0}, [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]),
        # nothing relevant at all: every metric 0 (empty_target_action = "neg")
        (3, [4, 5, 6], {4: 0.0}, [0.0] * 6),
        # two hits: k = 4, retrieved [7, 8, 9, 6], targets {8: 1, 6: 5, 2: 3}: rel@4 = [0, 1, 0, 5]
        (4, [7, 8, 9, 6], {8: 1.0, 6: 5.0, 2: 3.0},
         [(1 / l2(3) + 5 / l2(5)) / (5 / l2(2) + 3 / l2(3) + 1 / l2(4)), 2 / 3, 2 / 4, (1 / 2 + 2 / 4) / 2, 1.0, 1 / 2]),
    ]
    for k, got, tgt, want in cases:
        out = oretr.retrieval_metrics(np.array([got]), [tgt], k)[0]
        np.testing.assert_allclose(out, want, rtol=1e-12, atol=1e-12, err_msg=str((k, got, tgt)))


# ------------------------------------------------------------------ key order --
def _orderable(x: float) -> int:          # independent of mf_numerics.h: order-preserving map via struct packing
    import struct

    u = struct.unpack("<I", struct.pack("<f", x))[0]
    return (~u & 0xFFFFFFFF) if u & 0x80000000 else (u | 0x80000000)


def test_mining_key_order_is_the_reference_order_refined():
    """semi_hard_mining (xfmr_rec/losses.py:134-162) prefers semi-hard negatives (below the positive, closest first),
    then hard ones (at or above it, closest first); the 64-bit key must sort exactly like the tuple
    (class, closeness, column) -- restated here with Python tuples, not with the shared header."""
    g = np.random.default_rng(5)
    b, n = 7, 60
    lg = g.normal(size=(b, n)).astype(np.float32)
    lg[:, 10] = lg[:, 3]                       # exact ties: lowest column first
    lg[2, 20] = lg[2, 2]                       # Dm = 0 exactly: a HARD negative (not below the positive)
    lg[4, 30] = np.float32(lg[4, 4] - 1e-38)   # denormal gap
    mask = g.random((b, n)) > 0.2
    mask[np.arange(b), np.arange(b)] = False
    keys = chain.mining_keys(lg, mask)
    for i in range(b):
        cols = [j for j in range(n) if mask[i, j]]
        dm = {j: np.float32(lg[i, j]) - np.float32(lg[i, i]) for j in cols}
        want = sorted(cols, key=lambda j: (0, -float(dm[j]), j) if dm[j] < 0 else (1, float(dm[j]), j))
        got = sorted(cols, key=lambda j: -int(keys[i, j]))
        assert got == want, i
        assert all(int(keys[i, j]) == 0 for j in range(n) if not mask[i, j])
        # and the reference's own sort value orders the same way wherever it is not tied
        ref = {j: (dm[j] - min(dm.values())) if dm[j] < 0 else -dm[j] for j in cols}
        for a, c in zip(got, got[1:]):
            assert ref[a] >= ref[c]


def test_retrieval_key_order_is_score_desc_then_row_asc():
    g = np.random.default_rng(6)
    q = g.normal(size=(3, 16)).astype(np.float32)
    items = g.normal(size=(80, 16)).astype(np.float32)
    items[40] = items[4]
    items[41] = -items[4]
    items[70] = 0.0                            # score +0.0 ...
    q[1] = 0.0                                 # ... and a query whose scores are all zero
    sc = chain.scores(q, items)
    s, i = chain.topk(q, items, 80)
    for r in range(3):
        want = sorted(range(80), key=lambda j: (-_orderable(float(sc[r, j])), j))
        assert i[r].tolist() == want
        assert np.array_equal(s[r].view(np.uint32), sc[r][want].view(np.uint32))
