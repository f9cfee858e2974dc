"""Randomised cross-checks of the kernels (minutes of GPU time as a script; tests/test_gpu_stress.py runs a time-boxed slice).
  * top-k: the three retrieval paths (fp32 tiles, few-query scan, bf16 prefilter) must return the same bits on random
    shapes -- Q, N (incl. N % 32 != 0 and tiny N), d, k, exclusion lists, idx_base, duplicate rows, scaled norms;
  * the one-launch training step against the multi-kernel step (torch.equal) on random small shapes;
  * sparse update: the one-launch path against the oracle's coalesced row-Adam on random id multisets (Zipf, uniform,
    one dominant id, out-of-range ids), and bit-reproducibility of a repeated call;
  * mined losses: the split-bf16 candidate search against the fp32 streaming selection (the same mask bits, the same loss
    and gradient bits) on random shapes it serves -- Zipf item ids with rows looked up by id (hundreds of exact copies),
    logQ, large sigma, zero and negative targets, users that do not list their own positive, near-identical item rows.
    python tests/stress_gpu.py [seconds]      (test infrastructure: it checks the kernels against oracle/)"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import embed as oembed  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
lib = mf._lib.lib()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(int(os.environ.get("STRESS_SEED", "1")))


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def topk_case():
    d = [32, 64, 128, 256][ri(0, 3)]
    n = [ri(1, 40), ri(41, 3000), ri(3000, 70000), ri(8192, 200000)][ri(0, 3)]
    q = [ri(1, 4), ri(5, 40), ri(41, 300), ri(300, 1500)][ri(0, 3)]
    k = [1, ri(2, 20), 20, 64][ri(0, 3)]
    base = [0, ri(1, 1000)][ri(0, 1)]
    items = torch.randn(n, d, generator=g)
    qs = torch.randn(q, d, generator=g)
    if ri(0, 1):
        items = torch.nn.functional.normalize(items, dim=-1)
        qs = torch.nn.functional.normalize(qs, dim=-1)
    else:
        items = items * torch.exp(torch.randn(n, 1, generator=g))
    if n > 10 and ri(0, 1):
        items[ri(0, n - 1)] = items[ri(0, n - 1)]
    if ri(0, 3) == 0:
        qs[ri(0, q - 1)] = 0.0
    if n > 300 and ri(0, 2) == 0:
        lo = ri(0, n - 200)
        items[lo: lo + 150] = qs[0] + 0.01 * torch.randn(150, d, generator=g)
    excl = None
    if ri(0, 1):
        excl = [sorted(set((torch.randint(0, n, (ri(0, min(300, n)),), generator=g) + base).tolist())) for _ in range(q)]
    index = mf.retrieval.ItemIndex(items.to(dev), idx_base=base)
    ref = index.search(qs.to(dev), k, exclude=excl, path="tiles")
    paths = []
    if q <= index.SMALL_Q:
        paths.append("scan")
    if d >= 64:
        paths.append("bf16")
    for path in paths:
        got = index.search(qs.to(dev), k, exclude=excl, path=path)
        if not (torch.equal(got[1], ref[1]) and torch.equal(got[0].view(torch.int32), ref[0].view(torch.int32))):
            print(f"TOPK MISMATCH path={path} q={q} n={n} d={d} k={k} base={base} excl={excl is not None}", flush=True)
            return False
    return True


def update_case():
    d = [32, 64, 128, 256][ri(0, 3)]
    rows = [ri(1, 50), ri(51, 5000), ri(5000, 200000)][ri(0, 2)]
    n = [ri(1, 40), ri(41, 5000), ri(5000, 65536), ri(65537, 90000)][ri(0, 3)]
    kind = ri(0, 3)
    if kind == 0:
        idx = torch.randint(0, rows, (n,), generator=g)
    elif kind == 1:
        w = 1.0 / torch.arange(1, rows + 1, dtype=torch.float64)
        idx = torch.multinomial(w, n, replacement=True, generator=g)
    elif kind == 2:
        idx = torch.randint(0, rows, (n,), generator=g)
        idx[torch.rand(n, generator=g) < 0.7] = ri(0, rows - 1)
    else:
        idx = torch.randint(-3, rows + 3, (n,), generator=g)
    grad = torch.randn(n, d, generator=g)
    valid = (idx >= 0) & (idx < rows)
    table0 = torch.randn(rows, d, generator=g)
    table = table0.to(dev)
    em, ev = torch.zeros_like(table), torch.zeros_like(table)
    ws = mf._lib.workspace(lib.mf_update_ws_bytes(n, d), dev)
    gi, gg = idx.to(dev), grad.to(dev)
    ref, rm, rv = table0.clone(), torch.zeros_like(table0), torch.zeros_like(table0)
    for step in (1, 2):
        mf._lib.check(lib.mf_update_adam(table.data_ptr(), em.data_ptr(), ev.data_ptr(), rows, d, gi.data_ptr(), n, gg.data_ptr(), 0,
                                         step, None, 0.05, 0.9, 0.999, 1e-8, 0.01, ws.data_ptr(), ws.numel(), None))
        if valid.any():
            oembed.adam_update(ref, rm, rv, idx[valid], grad[valid], step=step, lr=0.05, weight_decay=0.01)
    # sums of many gradient rows in a different order than index_add; near-zero sums make Adam's sign-like first steps differ by lr
    diff = (table.cpu() - ref).abs()
    bad = diff > (2e-4 * ref.abs() + 2e-5)
    if bad.float().mean() > 1e-4:
        print(f"UPDATE MISMATCH rows={rows} n={n} d={d} kind={kind} frac_bad={bad.float().mean():.2e} max={diff.max():.3e}", flush=True)
        return False
    t2 = table0.to(dev)
    m2, v2 = torch.zeros_like(t2), torch.zeros_like(t2)
    for step in (1, 2):
        mf._lib.check(lib.mf_update_adam(t2.data_ptr(), m2.data_ptr(), v2.data_ptr(), rows, d, gi.data_ptr(), n, gg.data_ptr(), 0,
                                         step, None, 0.05, 0.9, 0.999, 1e-8, 0.01, ws.data_ptr(), ws.numel(), None))
    if not torch.equal(t2, table):
        print(f"UPDATE NOT REPRODUCIBLE rows={rows} n={n} d={d} kind={kind}", flush=True)
        return False
    return True


def loss_case():
    """random tile-ragged shapes through every loss class, dense and mined, against the oracle (the forward's loop is
    unrolled by two tiles: odd / even tile counts, every split length)"""
    import numpy as np

    from oracle import chain, losses as ol
    from tests import test_gpu_parity as tp

    d = [32, 64, 128, 256][ri(0, 3)]
    b = [ri(1, 40), ri(41, 300), ri(300, 700)][ri(0, 2)]
    n = b + [0, ri(1, 64), ri(64, 900)][ri(0, 2)]
    p = ri(1, 40)
    k = [0, 0, ri(1, 8), ri(9, 64), n + 5][ri(0, 4)]
    if 0 < k < n and k > 64:
        k = 64
    sigma, margin = [(1.0, 1.0), (3.0, 0.25), (0.5, 0.0)][ri(0, 2)]
    t = tp._random_case(b, n, d, p, seed=ri(0, 10**6))
    logq = None if ri(0, 1) else torch.log(torch.rand(n, generator=g) * 0.9 + 0.05)
    lg = chain.logits(t["u"].numpy(), t["v"].numpy(), t["target"].numpy(), sigma, None if logq is None else logq.numpy())
    for kind in ol.KINDS:
        u = t["u"].clone().requires_grad_()
        v = t["v"].clone().requires_grad_()
        want = ol.loss(kind, u, v, t["target"], item_idx=t["item_idx"], pos_idx=t["pos_idx"], num_negatives=k, sigma=sigma,
                       margin=margin, logq=logq, mining_logits=lg)
        want.backward()
        got, du, dv = tp._run_gpu(mf, kind, t, k, sigma, margin, logq)
        w = float(want.detach())
        if not np.isfinite(w):
            ok = got == w or (np.isnan(got) and np.isnan(w))
        else:
            bu = ~np.isclose(du, u.grad.numpy(), rtol=2e-4, atol=2e-5 * sigma)
            bv = ~np.isclose(dv, v.grad.numpy(), rtol=2e-4, atol=2e-5 * sigma)
            rows_off = int(bu.any(axis=1).sum() + bv.any(axis=1).sum())
            # the hinge-type losses have step-function gradients: an element whose argument is within rounding of the kink
            # (the oracle's logits are torch's, ours the fmaf chain's: 1e-7 apart) flips one (user, item) contribution --
            # one row of du and one of dv.  Counted, not failed, when the loss agrees and at most 4 such pairs exist.
            kink = kind in ("ContrastiveLoss", "AlignmentContrastiveLoss", "PairwiseHingeLoss") and 0 < rows_off <= 8
            if kink:
                global n_kink
                n_kink += 1
            ok = abs(got - w) <= 1e-4 * sigma * max(1.0, abs(w)) and (rows_off == 0 or kink)
        if not ok:
            print(f"LOSS MISMATCH kind={kind} b={b} n={n} d={d} p={p} k={k} sigma={sigma} logq={logq is not None} got={got} want={w}", flush=True)
            return False
    return True


def small_step_case():
    """the one-launch step (mf_step_small) against the multi-kernel step: torch.equal on the loss, both tables and the Adam
    moments after two steps, on random small shapes (B <= 128, N <= 256), every mined loss, padded / CSR / no positives,
    logQ, duplicates, SGD and row-Adam"""
    d = [32, 64, 128, 256][ri(0, 3)]
    b = [ri(1, 8), ri(9, 64), ri(65, 128)][ri(0, 2)]
    n = min(256, b + [0, ri(1, 32), b, ri(33, 128)][ri(0, 3)])
    if n < 2:
        n = 2
    k = min([1, ri(1, 8), ri(9, 64), 64][ri(0, 3)], n - 1)
    kind = ["ContrastiveLoss", "AlignmentContrastiveLoss", "InfomationNoiseContrastiveEstimationLoss",
            "MutualInformationNeuralEstimationLoss", "PairwiseHingeLoss", "PairwiseLogisticLoss"][ri(0, 5)]
    n_users, n_items = ri(max(2, b // 2), 400), ri(max(2, n // 2), 600)
    adam = ri(0, 1)
    mode = ri(0, 2)                      # padded / CSR / no positives
    use_logq = ri(0, 1)
    seed = ri(0, 10**6)

    def make():
        torch.manual_seed(seed)
        towers = mf.models.init_towers(mf.models.ModelConfig(num_users=n_users, num_items=n_items, hidden_size=d), device=dev)
        opt = mf.optim.RowAdam(towers.parameters(), lr=0.05) if adam else mf.optim.SparseSGD(towers.parameters(), lr=0.1, weight_decay=0.01)
        return towers, opt

    ta, oa = make()
    tb, ob = make()
    fn = getattr(mf.losses, kind)(num_negatives=k, sigma=[1.0, 1.7][ri(0, 1)], margin=[1.0, 0.3][ri(0, 1)])
    logq = (torch.rand(n_items, generator=g) - 0.5).to(dev) if use_logq else None
    lists = [torch.randint(0, n_items, (ri(0, 20),), generator=g) for _ in range(n_users)]
    off = torch.tensor([0] + list(torch.tensor([x.numel() for x in lists]).cumsum(0).tolist()), dtype=torch.int64).to(dev)
    flat = (torch.cat(lists) if sum(x.numel() for x in lists) else torch.zeros(1, dtype=torch.int64)).to(dev)
    fused = mf.fused.FusedSmallStep(tb, ob, fn, logq_table=logq)
    one = torch.ones((), device=dev)
    tag = f"kind={kind} b={b} n={n} d={d} k={k} adam={adam} mode={mode} logq={use_logq} users={n_users} items={n_items}"
    for step in range(2):
        user = torch.randint(0, n_users, (b,), generator=g)
        item = torch.randint(0, n_items, (n,), generator=g)
        target = torch.randint(-1, 6, (b,), generator=g)
        batch = {"user": user.to(dev), "item": item.to(dev), "target": target.to(dev)}
        if mode == 0:
            pos = torch.randint(0, n_items, (b, ri(1, 70)), generator=g)
            pos[:, 0] = item[:b]
            batch["pos"] = pos.to(dev)
        elif mode == 1:
            batch["pos_csr"] = (batch["user"], off, flat)
        want = fn(ta["user"](batch["user"]), ta["item"](batch["item"]), batch["target"], item_idx=batch["item"], pos_idx=batch.get("pos"),
                  logq_table=logq, pos_csr=batch.get("pos_csr"))
        want.backward(one)
        oa.step()
        oa.zero_grad(set_to_none=True)
        got = fused(batch)
        if fused.fallback_steps:
            print(f"SMALL STEP FELL BACK {tag}", flush=True)
            return False
        same = torch.equal(got, want.detach()) or (torch.isnan(got) and torch.isnan(want)) or (torch.isinf(got) and torch.isinf(want) and (got > 0) == (want > 0))
        for name in ("user", "item"):
            same = same and torch.equal(ta[name].weight, tb[name].weight)
        if adam:
            for pa, pb in zip(oa.param_groups[0]["params"], ob.param_groups[0]["params"]):
                same = same and torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"]) and torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"])
        if not same:
            print(f"SMALL STEP MISMATCH step={step} {tag} got={float(got)} want={float(want)}", flush=True)
            return False
    return True


def mining_case():
    from tests import test_gpu_parity as tp

    d = [64, 128][ri(0, 1)]
    b = [ri(256, 600), ri(600, 1500)][ri(0, 1)]
    n = max(b, [ri(2048, 3000), ri(3000, 7000)][ri(0, 1)])
    k = [ri(1, 4), ri(5, 16), ri(17, 32)][ri(0, 2)]
    sigma = [1.0, 0.3, 30.0, 1000.0][ri(0, 3)]
    n_items = [n // 8, n // 2, 4 * n][ri(0, 2)]
    t = tp._random_case(b, n, d, ri(1, 12), seed=ri(0, 10**6), n_items=n_items)
    flavour = ri(0, 3)
    if flavour >= 1:                                          # rows by id, Zipf popularity: exact copies
        w = 1.0 / torch.arange(1, n_items + 1, dtype=torch.float64)
        ids = torch.multinomial(w, n, replacement=True, generator=g) + 1
        table = tp._unit(n_items + 1, d, g)
        if flavour == 3:                                      # ... and nearly equal rows under different ids
            table[1:n_items // 2] = table[1] + 1e-6 * torch.randn(n_items // 2 - 1, d, generator=g)
        t["item_idx"], t["v"] = ids, table[ids].clone()
        t["pos_idx"] = torch.randint(0, n_items + 1, t["pos_idx"].shape, generator=g)
        t["pos_idx"][:, 0] = ids[:b]
        if flavour == 2:
            t["pos_idx"][::3, 0] = 0                          # these users do not list their own positive
    logq = None if ri(0, 1) else torch.log(torch.rand(n, generator=g) * 0.9 + 0.05)
    kind = ["PairwiseHingeLoss", "PairwiseLogisticLoss", "InfomationNoiseContrastiveEstimationLoss"][ri(0, 2)]
    res = []
    for mode in (0, 2):                                       # (2: the prefilter wherever it can serve)
        lib.mf_set_mining_prefilter(mode)
        mask = mf.losses.negative_mask(t["u"].to(dev), t["v"].to(dev), t["target"].to(dev), item_idx=t["item_idx"].to(dev),
                                       pos_idx=t["pos_idx"].to(dev), num_negatives=k, sigma=sigma).cpu()
        got, du, dv = tp._run_gpu(mf, kind, t, k, sigma, 0.5, logq)
        res.append((mask, got, torch.from_numpy(du), torch.from_numpy(dv)))
    lib.mf_set_mining_prefilter(1)
    import gc
    gc.collect()                                              # (the autograd contexts hold workspaces of hundreds of MB in reference cycles)
    same = torch.equal(res[0][0], res[1][0]) and (res[0][1] == res[1][1] or (res[0][1] != res[0][1] and res[1][1] != res[1][1])) \
        and torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
    if not same:
        print(f"MINING PREFILTER MISMATCH b={b} n={n} d={d} k={k} sigma={sigma} flavour={flavour} n_items={n_items} logq={logq is not None} "
              f"kind={kind} mask rows off {int((res[0][0] != res[1][0]).any(1).sum())} loss {res[0][1]} vs {res[1][1]}", flush=True)
    return same


n_kink = 0
CASES = {"topk": topk_case, "update": update_case, "loss": loss_case, "small": small_step_case, "mining": mining_case}


def run(budget: float, only: str | None = None, seed: int | None = None) -> tuple[int, int]:
    """Round-robin over the case generators for ``budget`` seconds; (cases ok, cases failed)."""
    if seed is not None:
        g.manual_seed(seed)
    fns = (CASES[only],) if only else tuple(CASES.values())
    t0, n_ok, n_bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        for fn in fns:
            ok = fn()
            n_ok += ok
            n_bad += not ok
    print(f"stress: {n_ok} cases ok, {n_bad} failed in {time.time() - t0:.0f} s ({n_kink} hinge-kink flips tolerated)", flush=True)
    return n_ok, n_bad


if __name__ == "__main__":
    _, bad = run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, os.environ.get("STRESS_ONLY"))
    sys.exit(1 if bad else 0)
