"""GPU parity tests: the HIP path (through the Python surface -> ctypes -> C ABI of
libmf_hip.so) against the CPU oracle on identical seeded inputs and against the golden
vectors derived from the reference.  Bars: scores / masks / top-k bit-exact; loss and
gradients within 1e-4 (fp32, sigma = 1; scaled by sigma otherwise)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import chain, embed as oembed, losses as ol, retrieval as oretr
from tests import _golden_util as gu
from tests.conftest import GOLDEN, golden_files

pytestmark = pytest.mark.gpu
SIGMA_MARGIN = ((1.0, 1.0), (2.0, 0.5), (1.0, 0.0))
DEV = "cuda:0"


def _unit(n, d, gen):
    return torch.nn.functional.normalize(torch.randn(n, d, generator=gen), dim=-1)


# ----------------------------------------------------------------- score engine ---
@pytest.mark.parametrize("d", [32, 64, 128, 256])
def test_mfma_scores_equal_fmaf_chain_bitwise(mf, d):
    g = torch.Generator().manual_seed(d)
    u, v = torch.randn(70, d, generator=g), torch.randn(133, d, generator=g)
    out = torch.empty(70, 133, device=DEV)
    ud, vd = u.to(DEV), v.to(DEV)
    mf._lib.check(mf._lib.lib().mf_scores(ud.data_ptr(), 70, vd.data_ptr(), 133, d, out.data_ptr(), None))
    want = chain.scores(u.numpy(), v.numpy())
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_dpp_lane_exchanges_equal_the_shuffles_they_replace(mf):
    """mf_common.h's DPP / v_permlane swaps against `__shfl_xor`: butterfly sums of every width, the 64-bit wave maximum,
    every single exchange -- on random bit patterns (floats of every magnitude and sign; no NaN INPUTS: which payload the sum
    of two NaNs keeps depends on the operand order, which the swaps do not preserve in the upper half-wave)."""
    import ctypes
    fn = mf._lib.lib().mf_probe_lane_ops
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    g = torch.Generator().manual_seed(5)
    waves = 4096
    bits = torch.randint(-2**31, 2**31, (waves, 128), generator=g, dtype=torch.int64).to(torch.int32)
    finite = torch.randn(waves // 2, 128, generator=g).mul(torch.logspace(-20, 20, 128)[None, :]).view(torch.int32)
    bits &= ~(1 << 23)                           # (exponent never all ones: overflow to inf / inf - inf still happen in the sums)
    bits[: waves // 2] = finite                  # half the waves: ordinary floats over forty decades
    bad = torch.zeros(16, dtype=torch.int32, device=DEV)
    mf._lib.check(fn(bits.to(DEV).data_ptr(), waves, bad.data_ptr(), None))
    assert bad.cpu().tolist() == [0] * 16          # (mismatching lanes per operation, in the kernel's order)


def test_sqnorm_bitwise(mf):
    x = torch.randn(300, 128)
    out = torch.empty(300, device=DEV)
    xd = x.to(DEV)
    mf._lib.check(mf._lib.lib().mf_row_sqnorm(xd.data_ptr(), 300, 128, out.data_ptr(), None))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), chain.sqnorm(x.numpy()).view(np.uint32))


@pytest.mark.parametrize("n", [1, 5, 2048, 5000])
def test_sort_keys_is_stable_sort(mf, n):
    g = torch.Generator().manual_seed(n)
    keys = torch.randint(0, max(2, n // 3), (n,), generator=g)
    kd = keys.to(DEV)
    lib = mf._lib.lib()
    perm = torch.empty(n, dtype=torch.int32, device=DEV)
    sk = torch.empty(n, dtype=torch.int64, device=DEV)
    ws = mf._lib.workspace(lib.mf_sort_ws_bytes(n), DEV)
    mf._lib.check(lib.mf_sort_keys(kd.data_ptr(), n, perm.data_ptr(), sk.data_ptr(), ws.data_ptr(), ws.numel(), None))
    want = torch.argsort(keys, stable=True)
    assert torch.equal(perm.cpu().long(), want)
    assert torch.equal(sk.cpu(), keys[want])


# ------------------------------------------------------------------------ losses ---
def _run_gpu(mf, kind, t, k, sigma, margin, logq=None):
    u = t["u"].to(DEV).requires_grad_()
    v = t["v"].to(DEV).requires_grad_()
    fn = getattr(mf.losses, kind)(num_negatives=k, sigma=sigma, margin=margin)
    val = fn(u, v, t["target"].to(DEV), item_idx=t["item_idx"].to(DEV), pos_idx=t["pos_idx"].to(DEV),
             logq=None if logq is None else logq.to(DEV))
    val.backward()
    return float(val.detach().cpu()), u.grad.cpu().numpy(), v.grad.cpu().numpy()


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.stem)
def test_losses_match_reference_golden(mf, path):
    """Values produced by the reference's own xfmr_rec/losses.py (tests/golden)."""
    z = np.load(path)
    t = {k: torch.from_numpy(z[k]) for k in ("u", "v", "target", "item_idx", "pos_idx")}
    n = t["v"].shape[0]
    gs = int(z["gstride"])
    for ki, kind in enumerate(ol.KINDS):
        for k in (0, 4, n):
            for smi, (sigma, margin) in enumerate(SIGMA_MARGIN):
                tag = f"{ki}_{k}_{smi}"
                want = float(z[f"loss_{tag}"])
                got, du, dv = _run_gpu(mf, kind, t, k, sigma, margin)
                if not np.isfinite(want):
                    assert got == want, (path.stem, tag, got, want)   # MINE, no valid negative: -inf
                    continue
                assert abs(got - want) <= 1e-4 * sigma * max(1.0, abs(want)), (path.stem, tag, got, want)
                np.testing.assert_allclose(du[::gs], z[f"du_{tag}"], rtol=2e-4, atol=2e-5 * sigma, err_msg=tag)
                np.testing.assert_allclose(dv[::gs], z[f"dv_{tag}"], rtol=2e-4, atol=2e-5 * sigma, err_msg=tag)


@pytest.mark.parametrize("path", sorted(GOLDEN.glob("wide_*.npz")), ids=lambda p: p.stem)
def test_losses_match_reference_over_the_tuners_range(mf, path):
    """The ends of the range the reference's own tuners draw from (xfmr_rec/ray.py:147-149, flaml.py:73-79): sigma = 30 and
    1000 (exp underflows on every off-diagonal term), margin = -0.5, num_negatives = 1 / 4 / 32 -- against values the
    reference itself produced (tests/golden/make_golden.py).  Loss within rel 1e-5 + the effect of a 2e-6 * sigma logit error
    (inside the north-star's 1e-4 * sigma); mined masks identical up to near-ties of the reference's own sort key;
    gradients row-relative 1e-4 + 5e-6 * sigma, rows decided by a hinge kink / a near-tie left out (tests/_golden_util.py)."""
    z = np.load(path)
    t = {k: torch.from_numpy(z[k]) for k in ("u", "v", "target", "item_idx", "pos_idx")}
    b, n = t["u"].shape[0], t["v"].shape[0]
    gs = int(z["gstride"])
    skipped = total = 0
    for smi, (sigma, margin) in enumerate(z["sigma_margin"].tolist()):
        for k in z["ks"].tolist():
            got_mask = mf.losses.negative_mask(t["u"].to(DEV), t["v"].to(DEV), t["target"].to(DEV), item_idx=t["item_idx"].to(DEV),
                                               pos_idx=t["pos_idx"].to(DEV), num_negatives=k, sigma=sigma).cpu().numpy()
            gu.assert_masks_equal_up_to_ties(z, got_mask, k, smi, (path.stem, smi, k), tol=1e-5 * sigma)
            for ki, kind in enumerate(ol.KINDS):
                tag = f"{ki}_{k}_{smi}"
                want = float(z[f"loss_{tag}"])
                got, du, dv = _run_gpu(mf, kind, t, k, sigma, margin)
                assert abs(got - want) <= gu.loss_tolerance(want, sigma, z["target"]), (path.stem, tag, got, want)
                assert abs(got - want) <= 1e-4 * sigma * max(1.0, abs(want))                       # the north-star's bar
                rows, cols = gu.undecided(z, ki, k, smi, sigma, margin, got_mask)
                ru, rv = gu.keep(b, rows, gs), gu.keep(n, cols, gs)
                skipped += (len(range(0, b, gs)) - len(ru)) + (len(range(0, n, gs)) - len(rv))
                total += len(range(0, b, gs)) + len(range(0, n, gs))
                gu.assert_grads_close(du[::gs][ru], z[f"du_{tag}"][ru], sigma, (path.stem, "du", tag))
                gu.assert_grads_close(dv[::gs][rv], z[f"dv_{tag}"][rv], sigma, (path.stem, "dv", tag))
    assert skipped <= 0.05 * total, (skipped, total)


@pytest.mark.parametrize("kind", ["MutualInformationNeuralEstimationLoss", "InfomationNoiseContrastiveEstimationLoss", "PairwiseLogisticLoss"])
def test_mined_dv_sum_at_its_bound_does_not_wrap(mf, kind):
    """The mined dV accumulator is 64-bit fixed point (csrc/mf_loss_math.h ``dv_fix_of``).  Worst case of the reference's own
    range (xfmr_rec/ray.py:147-149): sigma = 1000, |target| = 5, and ONE column mined by every row of B = 8192, every term
    of the same sign and |u - v| = 2: the column's gradient is ~8e7 > 2^23, where the fixed 2^-40 unit of rounds 1-3 wrapped
    an int64 silently.  The unit now follows the batch: the column must match the oracle (run on the 8193
    columns that matter -- the others are never mined), never wrap."""
    b, n, d, sigma = 8192, 16384, 32, 1000.0
    g = torch.Generator().manual_seed(11)
    c = torch.zeros(d)
    c[0] = 1.0
    u = torch.nn.functional.normalize(c[None, :] + 0.01 * torch.randn(b, d, generator=g), dim=-1)
    v = torch.empty(n, d)
    v[:b] = -u                                   # the positives: as far as a unit vector gets (targets are negative: far = high logit)
    v[b] = -c                                    # THE column: just below every row's positive -> the closest semi-hard negative of all of them
    v[b + 1:] = torch.nn.functional.normalize(c[None, :] + 0.3 * torch.randn(n - b - 1, d, generator=g), dim=-1)      # far below
    target = torch.full((b,), -5, dtype=torch.int64)
    item_idx = torch.arange(n, dtype=torch.int64) + 100
    item_idx[:b] = 7                             # one id for every positive: the other rows' positives are accidental hits, never mined
    fn = getattr(mf.losses, kind)(num_negatives=1, sigma=sigma, margin=1.0)
    ud, vd = u.to(DEV).requires_grad_(), v.to(DEV).requires_grad_()
    mask = mf.losses.negative_mask(ud.detach(), vd.detach(), target.to(DEV), item_idx=item_idx.to(DEV), pos_idx=None, num_negatives=1, sigma=sigma)
    assert bool(mask[:, b].all()) and int(mask.sum()) == b                  # every row mined column b and nothing else
    val = fn(ud, vd, target.to(DEV), item_idx=item_idx.to(DEV), pos_idx=None)
    val.backward()
    # the oracle in fp64: the construction cancels hard along the common direction (two logits near 2000 that differ by ~1.5,
    # gradient terms of ~1e3 that leave ~1) -- its fp32 evaluation is itself 6 % off on that component
    uo, vo = u.double().requires_grad_(), v[: b + 1].double().requires_grad_()
    want = ol.loss(kind, uo, vo, target, item_idx=item_idx[: b + 1], pos_idx=None, num_negatives=1, sigma=sigma, margin=1.0)
    want.backward()
    assert abs(float(val) - float(want)) <= gu.loss_tolerance(float(want), sigma, target.numpy())
    dv = vd.grad.cpu().numpy()
    col = vo.grad[b].numpy()
    assert np.abs(col).max() > 2.0**23, float(np.abs(col).max())             # the case IS beyond the old accumulator's range
    assert not dv[b + 1:].any()
    gu.assert_grads_close(dv[b:b + 1], col[None, :], sigma, (kind, "the column"))
    gu.assert_grads_close(dv[:b:64], vo.grad[:b:64].numpy(), sigma, (kind, "dv"))
    gu.assert_grads_close(ud.grad.cpu().numpy()[::64], uo.grad.numpy()[::64], sigma, (kind, "du"))


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.stem)
def test_masks_bit_exact_vs_oracle(mf, path):
    z = np.load(path)
    t = {k: torch.from_numpy(z[k]) for k in ("u", "v", "target", "item_idx", "pos_idx")}
    b, n = t["u"].shape[0], t["v"].shape[0]
    for sigma in (1.0, 2.0):
        lg = torch.from_numpy(chain.logits(t["u"].numpy(), t["v"].numpy(), t["target"].numpy(), sigma))
        base = ol.negative_masks(t["item_idx"], t["pos_idx"], b)
        for k in (0, 4, n):
            want = ol.semi_hard_mining(lg, base.clone(), k)
            got = mf.losses.negative_mask(t["u"].to(DEV), t["v"].to(DEV), t["target"].to(DEV),
                                          item_idx=t["item_idx"].to(DEV), pos_idx=t["pos_idx"].to(DEV),
                                          num_negatives=k, sigma=sigma).cpu()
            assert torch.equal(got, want), (path.stem, sigma, k, (got != want).sum().item())


@pytest.mark.parametrize("cfg", [(300, 4096, 64, 4), (130, 2500, 128, 24), (64, 2100, 32, 64)], ids=lambda c: "x".join(map(str, c)))
def test_mined_masks_bit_exact_long_item_axis(mf, cfg):
    """Semi-hard mining where the item axis is long enough for the seeding pass and several chunks per row
    (the golden / ragged cases above are shorter): the mined mask is bit-exact, duplicates and hits included."""
    b, n, d, k = cfg
    t = _random_case(b, n, d, 6, seed=sum(cfg), n_items=n // 3)
    lg = torch.from_numpy(chain.logits(t["u"].numpy(), t["v"].numpy(), t["target"].numpy(), 1.0))
    want = ol.semi_hard_mining(lg, ol.negative_masks(t["item_idx"], t["pos_idx"], b), k)
    got = mf.losses.negative_mask(t["u"].to(DEV), t["v"].to(DEV), t["target"].to(DEV), item_idx=t["item_idx"].to(DEV),
                                  pos_idx=t["pos_idx"].to(DEV), num_negatives=k, sigma=1.0).cpu()
    assert torch.equal(got, want), int((got != want).sum())


def _random_case(b, n, d, p, seed, n_items=None):
    g = torch.Generator().manual_seed(seed)
    n_items = n_items or max(n // 2, 4)
    t = {
        "u": _unit(b, d, g), "v": _unit(n, d, g),
        "target": torch.randint(-2, 6, (b,), generator=g),
        "item_idx": torch.randint(1, n_items + 1, (n,), generator=g),
        "pos_idx": torch.randint(0, n_items + 1, (b, p), generator=g),
    }
    t["pos_idx"][:, 0] = t["item_idx"][:b]
    return t


@pytest.mark.parametrize("shape", [(200, 400, 64, 20), (33, 95, 32, 5), (64, 64, 128, 1), (100, 260, 256, 40)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("k", [0, 4, 24, 40])
def test_losses_match_oracle_ragged_shapes(mf, shape, k):
    """Tile-ragged shapes, every loss class, dense and mined, with and without logQ."""
    b, n, d, p = shape
    t = _random_case(b, n, d, p, seed=sum(shape) + k)
    g = torch.Generator().manual_seed(7)
    for logq in (None, torch.log(torch.rand(n, generator=g) * 0.9 + 0.05)):
        for sigma, margin in ((1.0, 1.0), (3.0, 0.25)):
            lg = chain.logits(t["u"].numpy(), t["v"].numpy(), t["target"].numpy(), sigma,
                              None if logq is None else logq.numpy())
            for kind in ol.KINDS:
                u = t["u"].clone().requires_grad_()
                v = t["v"].clone().requires_grad_()
                want = ol.loss(kind, u, v, t["target"], item_idx=t["item_idx"], pos_idx=t["pos_idx"], num_negatives=k,
                               sigma=sigma, margin=margin, logq=logq, mining_logits=lg)
                want.backward()
                got, du, dv = _run_gpu(mf, kind, t, k, sigma, margin, logq)
                w = float(want.detach())
                if not np.isfinite(w):
                    assert got == w or (np.isnan(got) and np.isnan(w)), (kind, got, w)
                    continue
                assert abs(got - w) <= 1e-4 * sigma * max(1.0, abs(w)), (kind, k, sigma, got, w)
                np.testing.assert_allclose(du, u.grad.numpy(), rtol=2e-4, atol=2e-5 * sigma, err_msg=kind)
                np.testing.assert_allclose(dv, v.grad.numpy(), rtol=2e-4, atol=2e-5 * sigma, err_msg=kind)


@pytest.mark.parametrize("k", [0, 4])
def test_prepared_masks_give_the_same_loss_and_gradients(mf, k):
    """Masks built ahead on the side stream (prepare_masks) == masks built inside the forward, bit for bit."""
    t = _random_case(150, 420, 64, 9, seed=77)
    fn = mf.losses.PairwiseHingeLoss(num_negatives=k)
    dev = {name: v.to(DEV) for name, v in t.items()}
    outs = []
    for use in (False, True):
        u, v = dev["u"].clone().requires_grad_(), dev["v"].clone().requires_grad_()
        prep = fn.prepare_masks(dev["item_idx"], dev["pos_idx"], batch_size=150, embedding_dim=64) if use else None
        loss = fn(u, v, dev["target"], item_idx=dev["item_idx"], pos_idx=dev["pos_idx"], prepared=prep)
        loss.backward()
        outs.append((loss.detach().clone(), u.grad.clone(), v.grad.clone()))
    # dense and mined alike: bit for bit (the mined dV is summed as exact fixed-point integers)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    with pytest.raises(ValueError, match="another batch"):
        other = dev["item_idx"].clone()
        fn(dev["u"], dev["v"], dev["target"], item_idx=other, pos_idx=dev["pos_idx"],
           prepared=fn.prepare_masks(dev["item_idx"], dev["pos_idx"], batch_size=150, embedding_dim=64))


def test_fused_losses_equal_individual(mf):
    t = _random_case(96, 192, 64, 8, seed=5)
    dev = {k: x.to(DEV) for k, x in t.items()}
    fused = mf.losses.fused_losses(dev["u"], dev["v"], dev["target"], item_idx=dev["item_idx"], pos_idx=dev["pos_idx"],
                                   num_negatives=4, sigma=1.0, margin=1.0)
    for kind in ol.KINDS:
        single = getattr(mf.losses, kind)(num_negatives=4)(dev["u"], dev["v"], dev["target"], item_idx=dev["item_idx"],
                                                            pos_idx=dev["pos_idx"])
        assert float(fused[kind]) == float(single), kind


@pytest.mark.parametrize("k", [0, 4])
def test_int64_targets_logq_table_and_named_train_loss(mf, k):
    """The reference hands int64 ratings (data/lightning.py:72-76): taken as they are == their fp32 copy, bit for bit.
    A logQ TABLE looked up by item_idx inside the kernel == the per-column values gathered by the caller (ids outside
    the table count as 0).  fused_losses(train_loss=...) back-propagates exactly what the single-loss module does."""
    t = _random_case(130, 300, 64, 7, seed=91)
    dev = {name: v.to(DEV) for name, v in t.items()}
    table = torch.log(torch.rand(200, generator=torch.Generator().manual_seed(3)) * 0.9 + 0.05).to(DEV)   # ids run to ~150
    short = table[:100]                                     # ids >= 100 fall outside: logq = 0 there
    fn = mf.losses.InfomationNoiseContrastiveEstimationLoss(num_negatives=k)

    def run(target, **kw):
        u, v = dev["u"].clone().requires_grad_(), dev["v"].clone().requires_grad_()
        loss = fn(u, v, target, item_idx=dev["item_idx"], pos_idx=dev["pos_idx"], **kw)
        loss.backward()
        return loss.detach(), u.grad, v.grad

    def same(a, b):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])

    same(run(dev["target"]), run(dev["target"].float()))
    same(run(dev["target"], logq_table=table), run(dev["target"], logq=table[dev["item_idx"]]))
    ids = dev["item_idx"]
    gathered = torch.where(ids < 100, short[ids.clamp(max=99)], torch.zeros((), device=DEV))
    same(run(dev["target"], logq_table=short), run(dev["target"], logq=gathered))
    with pytest.raises(ValueError, match="not both"):
        fn(dev["u"], dev["v"], dev["target"], item_idx=ids, pos_idx=dev["pos_idx"], logq=gathered, logq_table=table)

    # fused: every loss evaluated, one named as the trained one
    u, v = dev["u"].clone().requires_grad_(), dev["v"].clone().requires_grad_()
    vals = mf.losses.fused_losses(u, v, dev["target"], item_idx=ids, pos_idx=dev["pos_idx"], num_negatives=k,
                                  train_loss="InfomationNoiseContrastiveEstimationLoss")
    vals["InfomationNoiseContrastiveEstimationLoss"].backward()
    same((vals["InfomationNoiseContrastiveEstimationLoss"].detach(), u.grad, v.grad), run(dev["target"]))
    with pytest.raises(NotImplementedError, match="semi-hard mining supports"):
        mf.losses.fused_losses(dev["u"], dev["v"], dev["target"], item_idx=ids, pos_idx=dev["pos_idx"], num_negatives=100)


def test_check_inputs_errors(mf):
    fn = mf.losses.PairwiseHingeLoss()
    u, v = torch.randn(4, 32, device=DEV), torch.randn(8, 32, device=DEV)
    idx = torch.arange(8, device=DEV)
    with pytest.raises(ValueError, match="2 dimensions"):
        fn(u[0], v, torch.ones(4, device=DEV), item_idx=idx, pos_idx=None)
    with pytest.raises(ValueError, match="dimension 1"):
        fn(u, v[:, :16], torch.ones(4, device=DEV), item_idx=idx, pos_idx=None)
    with pytest.raises(ValueError, match="dimension 0"):
        fn(u, v, torch.ones(5, device=DEV), item_idx=idx, pos_idx=None)
    with pytest.raises(ValueError, match="dimension 0"):
        fn(v, u, torch.ones(8, device=DEV), item_idx=idx[:4], pos_idx=None)


# ------------------------------------------------------------ towers and updates ---
@pytest.mark.parametrize("d", [32, 64, 128, 256])
@pytest.mark.parametrize("normalize", [False, True])
def test_gather_rows(mf, d, normalize):
    g = torch.Generator().manual_seed(d)
    table = torch.randn(500, d, generator=g)
    idx = torch.randint(0, 500, (3, 77), generator=g)
    tower = mf.models.EmbeddingTower(500, d, normalize=normalize, device=DEV)
    with torch.no_grad():
        tower.weight.copy_(table.to(DEV))
    got = tower(idx.to(DEV)).detach().cpu()
    want = oembed.gather(table, idx, normalize)
    if normalize:
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    else:
        assert torch.equal(got, want)


@pytest.mark.parametrize("d", [32, 128])
@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_sparse_update_matches_oracle(mf, d, normalize, opt):
    g = torch.Generator().manual_seed(d + normalize)
    rows, n = 300, 700
    table0 = torch.randn(rows, d, generator=g)
    idx = torch.randint(0, 40, (n,), generator=g)          # heavy duplicates
    idx[:50] = torch.randint(0, rows, (50,), generator=g)
    idx[100:400] = 5                                       # one run far longer than a 32-row chunk
    tower = mf.models.EmbeddingTower(rows, d, normalize=normalize, device=DEV)
    with torch.no_grad():
        tower.weight.copy_(table0.to(DEV))
    optim = (mf.optim.SparseSGD(tower.parameters(), lr=0.1, weight_decay=0.01) if opt == "sgd"
             else mf.optim.RowAdam(tower.parameters(), lr=0.05, weight_decay=0.01))
    ref = table0.clone()
    m, v = torch.zeros_like(ref), torch.zeros_like(ref)
    for step in (1, 2, 3):
        gout = torch.randn(n, d, generator=g)
        out = tower(idx.to(DEV))
        out.backward(gout.to(DEV))
        optim.step()
        optim.zero_grad()
        graw = oembed.normalize_backward(ref[idx], gout) if normalize else gout
        if opt == "sgd":
            oembed.sgd_update(ref, idx, graw, 0.1, 0.01)
        else:
            oembed.adam_update(ref, m, v, idx, graw, step=step, lr=0.05, weight_decay=0.01)
        torch.testing.assert_close(tower.weight.detach().cpu(), ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("case", ["zipf", "one_row_overflows_its_bucket", "legacy_packed", "legacy_generic", "tiny"])
def test_sparse_update_paths(mf, case):
    """The one-launch update (batch-sized id lists: one workgroup per id bucket, LDS sort), its global-memory
    fallback for a bucket that does not fit LDS, and the multi-launch path of longer lists, all against the
    oracle's coalesced row-Adam; out-of-range ids are skipped and no other row is written."""
    lib = mf._lib.lib()
    g = torch.Generator().manual_seed(11)
    d = 64
    if case == "zipf":
        rows, n = 60_000, 16384
        w = 1.0 / torch.arange(1, rows + 1, dtype=torch.float64)
        idx = torch.multinomial(w, n, replacement=True, generator=g)
    elif case == "one_row_overflows_its_bucket":
        rows, n = 60_000, 16384
        idx = torch.randint(0, rows, (n,), generator=g)
        idx[torch.randperm(n, generator=g)[:9000]] = 4242          # > 8192 keys in one bucket
    elif case == "legacy_packed":
        rows, n = 1000, 70_000
        idx = torch.randint(0, rows, (n,), generator=g)
    elif case == "legacy_generic":
        rows, n = 5_000_000, 70_000
        idx = torch.randint(0, rows, (n,), generator=g)
        idx[100:180] = idx[7]
    else:
        rows, n = 50, 3
        idx = torch.tensor([7, 7, 9])
    if n > 3:
        idx[1], idx[2] = -1, rows + 9                     # out of range: ignored
    grad = torch.randn(n, d, generator=g)
    valid = (idx >= 0) & (idx < rows)
    touched = torch.unique(idx[valid])
    init = torch.randn(touched.numel(), d, generator=g)
    table = torch.zeros(rows, d, device=DEV)
    table[touched.to(DEV)] = init.to(DEV)
    em, ev = torch.zeros_like(table), torch.zeros_like(table)
    ws = mf._lib.workspace(lib.mf_update_ws_bytes(n, d), DEV)
    gi, gg = idx.to(DEV), grad.to(DEV)
    small, sm, sv = init.clone(), torch.zeros_like(init), torch.zeros_like(init)
    remap = torch.searchsorted(touched, idx[valid])
    for step in (1, 2):
        mf._lib.check(lib.mf_update_adam(table.data_ptr(), em.data_ptr(), ev.data_ptr(), rows, d, gi.data_ptr(), n, gg.data_ptr(), 0,
                                         step, None, 0.05, 0.9, 0.999, 1e-8, 0.01, ws.data_ptr(), ws.numel(), None))
        oembed.adam_update(small, sm, sv, remap, grad[valid], step=step, lr=0.05, weight_decay=0.01)
        # sums of thousands of gradient rows: the order of the additions differs from the oracle's index_add
        torch.testing.assert_close(table[touched.to(DEV)].cpu(), small, rtol=2e-4, atol=2e-5)
    assert int(torch.count_nonzero(table.abs().sum(dim=1))) == touched.numel()   # nothing else was written
    # the same call on the same inputs is bit-reproducible
    t2 = torch.zeros(rows, d, device=DEV)
    t2[touched.to(DEV)] = init.to(DEV)
    m2, v2 = torch.zeros_like(t2), torch.zeros_like(t2)
    for step in (1, 2):
        mf._lib.check(lib.mf_update_adam(t2.data_ptr(), m2.data_ptr(), v2.data_ptr(), rows, d, gi.data_ptr(), n, gg.data_ptr(), 0,
                                         step, None, 0.05, 0.9, 0.999, 1e-8, 0.01, ws.data_ptr(), ws.numel(), None))
    assert torch.equal(t2, table)


# --------------------------------------------------------------------- retrieval ---
@pytest.mark.parametrize("cfg", [(50, 5000, 64, 20), (1, 3883, 64, 20), (33, 1000, 128, 5), (7, 300, 32, 64), (40, 2500, 256, 32),
                                 (150, 20000, 128, 64), (300, 9000, 256, 40), (1030, 4100, 32, 1)],
                         ids=lambda c: "x".join(map(str, c)))
def test_topk_bit_exact(mf, cfg):
    nq, n, d, k = cfg
    g = torch.Generator().manual_seed(n)
    q, items = _unit(nq, d, g), _unit(n, d, g)
    items[10] = items[3]                                   # exact score ties -> lowest row first
    excl = [sorted(set(torch.randint(0, n, (int(torch.randint(0, 60, (1,), generator=g)),), generator=g).tolist()))
            for _ in range(nq)]
    index = mf.retrieval.ItemIndex(items.to(DEV))
    ws, wi = chain.topk(q.numpy(), items.numpy(), k, excl)
    paths = ("tiles", "scan") if nq <= index.SMALL_Q else ("tiles",)
    for path in paths + (("bf16",) if d >= 64 else ()):                       # every kernel, same bits
        s, i = index.search(q.to(DEV), k, exclude=excl, path=path)
        assert np.array_equal(i.cpu().numpy(), wi), path
        assert np.array_equal(s.cpu().numpy().view(np.uint32), ws.view(np.uint32)), path


@pytest.mark.parametrize("case", ["zero_query", "maxima_piled_in_one_lane_share", "k64_many_blocks"])
def test_topk_scan_path_hard_cases(mf, case):
    """The small-batch selection takes its bound from two block maxima per lane: inputs where that bound is weak (every
    score equal; the best rows all in blocks 0, 64, 128, ... -- one lane's share -- so that nearly every block passes
    it and the selection runs in rounds) must still give the tile engine's bits."""
    g = torch.Generator().manual_seed(3)
    n, d, k = 62423, 64, 64
    items = 0.01 * _unit(n, d, g)
    q = _unit(2, d, g)
    if case == "zero_query":
        q[0] = 0.0
    elif case == "maxima_piled_in_one_lane_share":
        for b in range(0, 976, 64):
            items[b * 64: b * 64 + 8] = q[0] * torch.linspace(0.5, 1.0, 8)[:, None] + 0.001 * torch.randn(8, d, generator=g)
    index = mf.retrieval.ItemIndex(items.to(DEV))
    st, it = index.search(q.to(DEV), k, path="tiles")
    ss, is_ = index.search(q.to(DEV), k, path="scan")
    assert torch.equal(it, is_)
    assert torch.equal(st.view(torch.int32), ss.view(torch.int32))


@pytest.mark.parametrize("case", ["random_1024", "zero_query_and_duplicates", "best_rows_in_one_group", "near_ties", "unnormalised",
                                  "few_rows", "d64_excl", "d256"])
def test_topk_bf16_prefilter_equals_tile_path(mf, case):
    """mf_topk_bf3 (two bf16 MFMA scans + exact rescoring of the rows above a rigorous bound) against the fp32 tile
    engine, bit for bit: random unit rows at the bench's shape, inputs that defeat the prefilter (a zero query --
    every row ties; thousands of copies of one row; the best rows packed into one 128-row group: all answered by the
    in-kernel exact scan), scores closer than the bf16 error, rows of very different norms, tiny catalogs,
    exclusions and idx_base."""
    g = torch.Generator().manual_seed(len(case))
    nq, n, d, k, base = 200, 20000, 128, 20, 0
    excl = None
    if case == "random_1024":
        nq, n = 1024, 62423
    elif case == "few_rows":
        nq, n, k = 70, 45, 20
    elif case == "d64_excl":
        nq, n, d, k, base = 300, 30000, 64, 64, 11
    elif case == "d256":
        nq, n, d, k = 100, 9000, 256, 10
    q, items = _unit(nq, d, g), _unit(n, d, g)
    if case == "zero_query_and_duplicates":
        q[3] = 0.0
        items[5000:9000] = items[17]                        # 4,000 copies: all tie with row 17
        q[4] = items[17]
    elif case == "best_rows_in_one_group":
        items[256:384] = q[0] + 0.05 * torch.randn(128, d, generator=g)      # the 128 best rows of query 0 share a group
    elif case == "near_ties":
        items[1000:1040] = q[1] + 1e-4 * torch.randn(40, d, generator=g)     # 40 scores within ~1e-5 of each other
    elif case == "unnormalised":
        items *= torch.exp(2.0 * torch.randn(n, 1, generator=g))
        q *= torch.exp(torch.randn(nq, 1, generator=g))
    elif case == "d64_excl":
        excl = [sorted(set((torch.randint(0, n, (int(torch.randint(0, 400, (1,), generator=g)),), generator=g) + base).tolist()))
                for _ in range(nq)]
    index = mf.retrieval.ItemIndex(items.to(DEV), idx_base=base)
    st, it = index.search(q.to(DEV), k, exclude=excl, path="tiles")
    sb, ib = index.search(q.to(DEV), k, exclude=excl, path="bf16")
    assert torch.equal(it, ib)
    assert torch.equal(st.view(torch.int32), sb.view(torch.int32))
    if case == "random_1024":                               # what "auto" picks at this shape
        sa, ia = index.search(q.to(DEV), k)
        assert torch.equal(ia, it) and torch.equal(sa.view(torch.int32), st.view(torch.int32))
    with pytest.raises(ValueError, match="at least 64"):
        mf.retrieval.ItemIndex(torch.zeros(100, 32, device=DEV)).search(torch.zeros(2, 32, device=DEV), 5, path="bf16")


@pytest.mark.parametrize("cfg", [(1, 62423, 128, 20), (1, 300000, 32, 20), (32, 70000, 64, 64), (5, 1000, 256, 7), (9, 64, 32, 64),
                                 (2, 1, 128, 3)], ids=lambda c: "x".join(map(str, c)))
def test_topk_scan_path_equals_tile_path(mf, cfg):
    """The small-batch matrix-vector scan (mf_topk_small: blocked catalog, one row per lane) against the MFMA tile
    engine on the same inputs: scores and rows bit for bit -- catalogs of one block, of several passes per workgroup,
    exclusions, ties, idx_base."""
    nq, n, d, k = cfg
    g = torch.Generator().manual_seed(n + nq)
    q, items = _unit(nq, d, g), _unit(n, d, g)
    if n > 10:
        items[10] = items[3]
    excl = [sorted(set(torch.randint(0, n, (int(torch.randint(0, 300, (1,), generator=g)),), generator=g).tolist()))
            for _ in range(nq)]
    excl = [[r + 7 for r in e] for e in excl]             # global rows: the index starts at row 7
    index = mf.retrieval.ItemIndex(items.to(DEV), idx_base=7)
    st, it = index.search(q.to(DEV), k, exclude=excl, path="tiles")
    ss, is_ = index.search(q.to(DEV), k, exclude=excl, path="scan")
    assert torch.equal(it, is_)
    assert torch.equal(st.view(torch.int32), ss.view(torch.int32))
    if n <= 70000:
        ws, wi = chain.topk(q.numpy(), items.numpy(), k, [[r - 7 for r in e] for e in excl])
        assert np.array_equal(np.where(is_.cpu().numpy() >= 0, is_.cpu().numpy() - 7, -1), wi)
    with pytest.raises(ValueError, match="at most"):
        index.search(torch.zeros(33, d, device=DEV), k, path="scan")


@pytest.mark.parametrize("d", [64, 256])
def test_topk_few_query_scan_with_long_exclusion_lists(mf, d):
    """The MFMA scan of the few-query path stages all queries' exclusion entries in LDS (6,144 entries); more than that --
    31 queries x ~450 ids -- takes its per-query list walks.  Both against the tile engine, bit for bit; rows past N in the
    last block, an id list that names rows outside the catalog."""
    n, nq, k = 5003, 31, 20
    g = torch.Generator().manual_seed(d)
    q, items = _unit(nq, d, g), _unit(n, d, g)
    index = mf.retrieval.ItemIndex(items.to(DEV))
    for per in (150, 450):                                            # 4,650 entries: staged (<= 8,192); 13,950: the fallback
        excl = [torch.randint(-5, n + 5, (per,), generator=g).tolist() for _ in range(nq)]
        st, it = index.search(q.to(DEV), k, exclude=excl, path="tiles")
        ss, is_ = index.search(q.to(DEV), k, exclude=excl, path="scan")
        assert torch.equal(it, is_), per
        assert torch.equal(st.view(torch.int32), ss.view(torch.int32)), per
        sa, ia = index.search(q.to(DEV), k, exclude=excl)             # "auto": the long lists go to the prefilter (d >= 64)
        assert torch.equal(it, ia) and torch.equal(st.view(torch.int32), sa.view(torch.int32)), per
        for row, ex in zip(is_.cpu().tolist(), excl):
            assert not set(row) & set(ex)


def test_topk_degenerate_inputs(mf):
    """All scores equal (zero queries): lowest rows win; fewer than k candidates: -1 padding."""
    items = torch.randn(400, 32)
    index = mf.retrieval.ItemIndex(items.to(DEV))
    small = mf.retrieval.ItemIndex(items[:6].to(DEV))
    for path in ("tiles", "scan"):
        s, i = index.search(torch.zeros(3, 32, device=DEV), 20, path=path)
        assert torch.equal(i.cpu(), torch.arange(20).repeat(3, 1)), path
        s, i = small.search(torch.randn(2, 32).to(DEV), 8, exclude=[[0, 1], []], path=path)
        assert (i[0, 4:] == -1).all() and (i[1, 6:] == -1).all() and torch.isinf(s[0, 4:]).all(), path


def test_topk_degenerate_inputs_long_catalog(mf):
    """Equal scores over a catalog long enough for the seeding pass and many chunks: every list overflows,
    the exact wave-cooperative path and its 64-bit floors decide, the lowest rows win; with exclusions."""
    n, k = 6000, 33
    items = torch.randn(n, 64)
    index = mf.retrieval.ItemIndex(items.to(DEV))
    excl = [[0, 2, 5], [], list(range(40)), [n - 1]]
    for path in ("tiles", "scan"):
        s, i = index.search(torch.zeros(4, 64, device=DEV), k, exclude=excl, path=path)
        for r, ex in enumerate(excl):
            want = [j for j in range(n) if j not in set(ex)][:k]
            assert i[r].cpu().tolist() == want, path
    # two distinct score levels: the k best are the rows of the upper level, in row order
    q = torch.zeros(1, 64)
    q[0, 0] = 1.0
    items2 = torch.zeros(n, 64)
    items2[::7, 0] = 1.0
    for path in ("tiles", "scan"):
        s, i = mf.retrieval.ItemIndex(items2.to(DEV)).search(q.to(DEV), k, path=path)
        assert i[0].cpu().tolist() == list(range(0, 7 * k, 7)) and bool((s[0] == 1.0).all()), path


def test_sharded_topk_merge_equals_full(mf):
    g = torch.Generator().manual_seed(11)
    q, items = _unit(64, 64, g), _unit(4001, 64, g)
    full_s, full_i = mf.retrieval.ItemIndex(items.to(DEV)).search(q.to(DEV), 20)
    bounds = np.linspace(0, 4001, 9).astype(int)
    ps, pi = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        s, i = mf.retrieval.ItemIndex(items[lo:hi].to(DEV), idx_base=int(lo)).search(q.to(DEV), 20)
        ps.append(s)
        pi.append(i)
    ms, mi = mf.retrieval.merge_topk(torch.stack(ps), torch.stack(pi), 20)
    assert torch.equal(mi, full_i) and torch.equal(ms, full_s)
    os_, oi = oretr.merge_topk(torch.stack(ps).cpu().numpy(), torch.stack(pi).cpu().numpy(), 20)
    assert np.array_equal(oi, mi.cpu().numpy())


def test_item_processor_search_surface(mf):
    cfg = mf.models.ModelConfig(num_users=50, num_items=200, hidden_size=32)
    towers = mf.models.init_towers(cfg, device=DEV)
    proc = mf.retrieval.ItemProcessor(item_ids=list(range(1000, 1200)))
    proc.get_index(towers["item"])
    emb = towers["user"](torch.tensor([3], device=DEV)).detach().cpu().numpy()
    df = proc.search(emb, exclude_item_ids=[1000, 1001], top_k=10)
    assert list(df.columns) == ["movie_rn", "movie_id", "score"] and len(df) == 10
    assert df["score"].is_monotonic_decreasing and not set(df["movie_id"]) & {1000, 1001}
    with pytest.raises(ValueError, match="must be intialised first"):
        mf.retrieval.ItemProcessor().search(emb)


def test_backward_can_be_repeated_and_fused_losses_backprop_together(mf):
    """Two losses of one fused forward receive gradient (two HIP backwards on the same workspace),
    and retain_graph backward twice gives identical gradients: the stashed logits stay intact."""
    t = _random_case(96, 192, 64, 8, seed=9)
    dev = {k: x.to(DEV) for k, x in t.items()}
    u = dev["u"].clone().requires_grad_()
    v = dev["v"].clone().requires_grad_()
    out = mf.losses.fused_losses(u, v, dev["target"], item_idx=dev["item_idx"], pos_idx=dev["pos_idx"])
    (out["InfomationNoiseContrastiveEstimationLoss"] + 0.5 * out["PairwiseLogisticLoss"]).backward()
    uo = t["u"].clone().requires_grad_()
    vo = t["v"].clone().requires_grad_()
    a = ol.loss("InfomationNoiseContrastiveEstimationLoss", uo, vo, t["target"], item_idx=t["item_idx"], pos_idx=t["pos_idx"])
    b = ol.loss("PairwiseLogisticLoss", uo, vo, t["target"], item_idx=t["item_idx"], pos_idx=t["pos_idx"])
    (a + 0.5 * b).backward()
    np.testing.assert_allclose(u.grad.cpu().numpy(), uo.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(v.grad.cpu().numpy(), vo.grad.numpy(), rtol=2e-4, atol=2e-5)
    u2 = dev["u"].clone().requires_grad_()
    loss = mf.losses.PairwiseHingeLoss()(u2, dev["v"], dev["target"], item_idx=dev["item_idx"], pos_idx=dev["pos_idx"])
    loss.backward(retain_graph=True)
    g1 = u2.grad.clone()
    u2.grad = None
    loss.backward()
    assert torch.equal(g1, u2.grad)


@pytest.mark.parametrize("kind", ["PairwiseHingeLoss", "InfomationNoiseContrastiveEstimationLoss"])
def test_mined_backward_is_bit_identical_on_repeat(mf, kind):
    """The reference's default configuration (PairwiseHingeLoss, num_negatives = 4, xfmr_rec/lightning.py:38-39) on a
    Zipf-like batch where popular items are mined by hundreds of users: many contributions per dv row, summed in
    whatever order the atomics land -- as exact integers, so five runs give the same bits (and match the oracle)."""
    g = torch.Generator().manual_seed(4)
    b, n, d = 700, 1400, 64
    t = {"u": _unit(b, d, g), "v": _unit(n, d, g), "target": torch.randint(1, 6, (b,), generator=g)}
    w = 1.0 / torch.arange(1, 41, dtype=torch.float64)
    t["item_idx"] = torch.multinomial(w, n, replacement=True, generator=g) + 1         # 40 distinct items: heavy duplicates
    t["pos_idx"] = torch.zeros(b, 1, dtype=torch.int64)
    t["pos_idx"][:, 0] = t["item_idx"][:b]
    dev = {name: x.to(DEV) for name, x in t.items()}
    fn = getattr(mf.losses, kind)(num_negatives=4)
    runs = []
    for _ in range(5):
        u, v = dev["u"].clone().requires_grad_(), dev["v"].clone().requires_grad_()
        fn(u, v, dev["target"], item_idx=dev["item_idx"], pos_idx=dev["pos_idx"]).backward()
        runs.append((u.grad.clone(), v.grad.clone()))
    for du, dv in runs[1:]:
        assert torch.equal(du, runs[0][0]) and torch.equal(dv, runs[0][1])
    uo, vo = t["u"].clone().requires_grad_(), t["v"].clone().requires_grad_()
    lg = chain.logits(t["u"].numpy(), t["v"].numpy(), t["target"].numpy(), 1.0, None)
    ol.loss(kind, uo, vo, t["target"], item_idx=t["item_idx"], pos_idx=t["pos_idx"], num_negatives=4, mining_logits=lg).backward()
    np.testing.assert_allclose(runs[0][1].cpu().numpy(), vo.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(runs[0][0].cpu().numpy(), uo.grad.numpy(), rtol=2e-4, atol=2e-5)


def test_mine_loss_row_without_valid_negatives(mf):
    """A row all of whose columns are hits has MINE = -inf like the reference (losses.py:242-244).  Its GRADIENT is
    where this path deviates on purpose: the reference back-propagates NaN through log(0) into every
    row of the batch it touches; here the empty softmax contributes nothing and the row keeps the finite gradient of
    its positive term -(-L_ii) -- so the other rows' gradients survive.  This test states that contract."""
    g = torch.Generator().manual_seed(11)
    u, v = _unit(2, 32, g), _unit(2, 32, g)
    item = torch.tensor([5, 5])                  # both columns carry user 0's and user 1's own item: every column is a hit
    target = torch.tensor([3, 2])
    ud, vd = u.to(DEV).requires_grad_(), v.to(DEV).requires_grad_()
    loss = mf.losses.MutualInformationNeuralEstimationLoss()(ud, vd, target.to(DEV), item_idx=item.to(DEV), pos_idx=None)
    assert float(loss) == float("-inf")
    loss.backward()
    assert torch.isfinite(ud.grad).all() and torch.isfinite(vd.grad).all()
    # d/du_i of  w_i * sigma * 0.5 * |u_i - v_i|^2  =  w_i (u_i - v_i)
    want = target.float()[:, None] * (u - v)
    torch.testing.assert_close(ud.grad.cpu(), want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(vd.grad.cpu(), -want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("path", golden_files()[:3], ids=lambda p: p.stem)
def test_public_mask_and_mining_helpers_match_reference(mf, path):
    """negative_masks / semi_hard_mining / hard_mining as methods, against the reference's own masks
    (golden) and the oracle, on the reference's own logits."""
    z = np.load(path)
    t = {k: torch.from_numpy(z[k]) for k in ("u", "v", "target", "item_idx", "pos_idx")}
    b, n = t["u"].shape[0], t["v"].shape[0]
    lg = torch.from_numpy(z["logits_0"])
    fn0 = mf.losses.PairwiseHingeLoss(num_negatives=0)
    base = fn0.negative_masks(lg.to(DEV), item_idx=t["item_idx"].to(DEV), pos_idx=t["pos_idx"].to(DEV)).cpu()
    want0 = np.unpackbits(z["mask_0_0"])[: b * n].reshape(b, n).astype(bool)
    assert np.array_equal(base.numpy(), want0)
    fn4 = mf.losses.PairwiseHingeLoss(num_negatives=4)
    semi = fn4.semi_hard_mining(lg.to(DEV), base.to(DEV)).cpu()
    assert torch.equal(semi, ol.semi_hard_mining(lg, base.clone(), 4))
    hard = fn4.hard_mining(lg.to(DEV), base.to(DEV)).cpu()
    assert torch.equal(hard, ol.hard_mining(lg, base.clone(), 4))
    # ... and the reference's own hard_mining output (fixture): the same set, or another pick among equal logits at the cut
    want_h = np.unpackbits(z["hard_4_0"])[: b * n].reshape(b, n).astype(bool)
    for i in np.nonzero((hard.numpy() != want_h).any(1))[0]:
        assert hard[i].sum() == want_h[i].sum()
        assert np.array_equal(np.sort(lg[i].numpy()[hard[i].numpy()]), np.sort(lg[i].numpy()[want_h[i]]))


def test_public_functions_and_per_loss_methods(mf):
    """squared_distance / weighted_mean (losses.py:9-23) and the per-loss methods (losses.py:164-246, :348-359) exist
    with the reference's signatures and values: squared_distance against the chain products bit for bit and against
    0.5 cdist^2 within 1e-6, its gradient against autograd of the formula; every per-loss method equals the class that
    upstream routes to it."""
    g = torch.Generator().manual_seed(3)
    q, c = _unit(37, 48, g), _unit(70, 48, g)                       # width 48: zero-padded to 64 inside
    qd, cd = q.to(DEV).requires_grad_(), c.to(DEV).requires_grad_()
    dist = mf.losses.squared_distance(qd, cd)
    want = 0.5 * torch.cdist(q, c) ** 2
    torch.testing.assert_close(dist.detach().cpu(), want, rtol=0, atol=1e-6)
    w = torch.rand(37, 70, generator=g)
    (dist * w.to(DEV)).sum().backward()
    qo, co = q.clone().requires_grad_(), c.clone().requires_grad_()
    (ol.half_sqdist(qo, co) * w).sum().backward()
    torch.testing.assert_close(qd.grad.cpu(), qo.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cd.grad.cpu(), co.grad, rtol=1e-5, atol=1e-6)
    m = torch.rand(5, 9, generator=g) > 0.5
    x = torch.randn(5, 9, generator=g)
    got = mf.losses.weighted_mean(x.to(DEV), m.to(DEV), dim=-1).cpu()
    torch.testing.assert_close(got, (x * m / (m.sum(-1, keepdim=True) + 1e-10)).sum(-1))
    t = _random_case(40, 80, 32, 5, seed=17)
    dev = {k: v.to(DEV) for k, v in t.items()}
    base = mf.losses.PairwiseHingeLoss(num_negatives=4, sigma=1.5, margin=0.3)
    kw = dict(item_idx=dev["item_idx"], pos_idx=dev["pos_idx"])
    for method, cls in (("contrastive_loss", "ContrastiveLoss"), ("infonce_loss", "InfomationNoiseContrastiveEstimationLoss"),
                        ("mine_loss", "MutualInformationNeuralEstimationLoss")):
        a = getattr(base, method)(dev["u"], dev["v"], dev["target"], **kw)
        b_ = getattr(mf.losses, cls)(num_negatives=4, sigma=1.5, margin=0.3)(dev["u"], dev["v"], dev["target"], **kw)
        assert torch.equal(a, b_), method
    a = base.alignment_loss(dev["u"], dev["v"], dev["target"])
    assert torch.equal(a, mf.losses.AlignmentLoss(sigma=1.5)(dev["u"], dev["v"], dev["target"], **kw))
    s = torch.randn(11, generator=g)
    torch.testing.assert_close(mf.losses.PairwiseHingeLoss().score_loss_fn(s), s.relu())
    torch.testing.assert_close(mf.losses.PairwiseLogisticLoss().score_loss_fn(s), -torch.nn.functional.logsigmoid(-s))


def test_sparse_update_generic_sort_path_and_out_of_range_ids(mf):
    """Ids outside the table are skipped (never written), the rest matches the oracle's coalesced update (a table
    too tall for a 32-bit id; the multi-launch sorts are covered by test_sparse_update_paths)."""
    lib = mf._lib.lib()
    rows, d, n = 5_000_000, 32, 700                      # (rows + 1) << 10 > 2^32
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, rows, (n,), generator=g)
    idx[100:180] = idx[7]                                # a run longer than two 32-row chunks
    idx[3], idx[4] = -1, rows + 9                        # out of range: ignored
    grad = torch.randn(n, d, generator=g)
    table = torch.zeros(rows, d, device=DEV)
    valid = (idx >= 0) & (idx < rows)
    touched = torch.unique(idx[valid])
    init = torch.randn(touched.numel(), d, generator=g)
    table[touched.to(DEV)] = init.to(DEV)
    ws = mf._lib.workspace(lib.mf_update_ws_bytes(n, d), DEV)
    gi, gg = idx.to(DEV), grad.to(DEV)
    mf._lib.check(lib.mf_update_sgd(table.data_ptr(), rows, d, gi.data_ptr(), n, gg.data_ptr(), 0, 0.1, 0.0,
                                    ws.data_ptr(), ws.numel(), None))
    small = torch.zeros(touched.numel(), d)
    small.copy_(init)
    remap = torch.searchsorted(touched, idx[valid])
    oembed.sgd_update(small, remap, grad[valid], 0.1, 0.0)
    torch.testing.assert_close(table[touched.to(DEV)].cpu(), small, rtol=2e-5, atol=2e-6)
    assert int(torch.count_nonzero(table.abs().sum(dim=1))) == touched.numel()   # nothing else was written


@pytest.mark.parametrize("cfg", [(70, 300, 0), (70, 300, 3), (257, 1000, 64), (40, 90, 130), (300, 2100, 17)],
                         ids=lambda c: "x".join(map(str, c)))
def test_negative_masks_random_ids_bit_exact(mf, cfg):
    """Hit masks on heavily duplicated random ids (Zipf-like), incl. no positives, ragged sizes, and a
    positive list too long for the LDS aggregation table (P = 130 -> global atomics path)."""
    b, n, p = cfg
    g = torch.Generator().manual_seed(b + n + p)
    item_idx = torch.randint(0, max(n // 6, 3), (n,), generator=g)         # many duplicate columns, id 0 included
    pos_idx = torch.randint(0, max(n // 6, 3) + 5, (b, p), generator=g) if p else None
    fn = mf.losses.PairwiseHingeLoss(num_negatives=0)
    got = fn.negative_masks(torch.zeros(b, n, device=DEV), item_idx=item_idx.to(DEV),
                            pos_idx=None if pos_idx is None else pos_idx.to(DEV)).cpu()
    want = ol.negative_masks(item_idx, pos_idx, b)
    assert torch.equal(got, want), int((got != want).sum())


def _csr_case(seed, b, n, n_users, n_items, lens, dup_user=None):
    """Ragged per-user lists (CSR over all users), a batch of users drawn with repetition, columns with duplicates."""
    g = torch.Generator().manual_seed(seed)
    lists = [torch.randperm(n_items - 1, generator=g)[: int(ln)] + 1 for ln in lens]
    off = torch.tensor([0] + list(np.cumsum([x.numel() for x in lists])), dtype=torch.int64)
    items = torch.cat(lists) if int(off[-1]) else torch.zeros(0, dtype=torch.int64)
    users = torch.randint(0, n_users, (b,), generator=g)
    if dup_user is not None:
        users[:3] = dup_user                                     # the heavy user several times in one 32-user group
        users[b // 2] = dup_user
    item_idx = torch.randint(1, n_items, (n,), generator=g)
    item_idx[n // 2: n // 2 + 5] = item_idx[0]                   # duplicate columns
    return lists, off, items, users, item_idx


@pytest.mark.parametrize("cfg", [(70, 300, 40_000, 30_000), (40, 6000, 6_001, 6_000), (33, 64, 500, 0)], ids=lambda c: "x".join(map(str, c)))
def test_csr_positives_masks_bit_exact(mf, cfg):
    """Positives as CSR lists read in place (mf_loss_fwd_csr) -- no [B, P] tensor: the mask equals the oracle's on the padded
    form of the same lists, incl. a 30,000-item user, empty lists, a user id outside the table (no positives),
    duplicates of the heavy user inside one 32-user group, and a group whose lists hit more first columns than the LDS
    table holds (6,000 > 4,096: the overflow goes out by global atomics).  The padded form through the same kernel too."""
    b, n, n_items, heavy = cfg
    n_users = 50
    g = torch.Generator().manual_seed(sum(cfg))
    lens = torch.randint(0, 40, (n_users,), generator=g)
    lens[3] = 0
    if heavy:
        lens[7] = heavy
        if n == 6000:
            lens[:] = heavy                                          # every user holds (nearly) every item: every column is a hit
    lists, off, items, users, item_idx = _csr_case(sum(cfg), b, n, n_users, n_items, lens.tolist(), dup_user=7)
    users[5] = n_users + 3                                           # outside pos_off: treated as an empty list
    longest = max(int(lens.max()), 1)
    pos_idx = torch.zeros(b, longest, dtype=torch.int64)
    for r, u in enumerate(users.tolist()):
        if u < n_users:
            pos_idx[r, : lists[u].numel()] = lists[u]
    want = ol.negative_masks(item_idx, pos_idx, b)
    u_, v_ = torch.zeros(b, 32, device=DEV), torch.zeros(n, 32, device=DEV)
    tgt = torch.ones(b, device=DEV)
    got = mf.losses.negative_mask(u_, v_, tgt, item_idx=item_idx.to(DEV), pos_csr=(users.to(DEV), off.to(DEV), items.to(DEV))).cpu()
    assert torch.equal(got, want), int((got != want).sum())
    got_p = mf.losses.negative_mask(u_, v_, tgt, item_idx=item_idx.to(DEV), pos_idx=pos_idx.to(DEV)).cpu()
    assert torch.equal(got_p, want), int((got_p != want).sum())
    if heavy:
        assert int((~want).sum()) > b                               # the lists do hit columns beyond the diagonal


# --------------------------------------------------------- hash / bloom towers (config 5) ---
@pytest.mark.parametrize("d", [32, 256])
@pytest.mark.parametrize("num_hashes", [1, 2, 4])
@pytest.mark.parametrize("normalize", [False, True])
def test_hash_tower_forward_matches_oracle(mf, d, num_hashes, normalize):
    g = torch.Generator().manual_seed(d + num_hashes)
    table = torch.randn(997, d, generator=g)
    idx = torch.randint(0, 10**9, (5, 41), generator=g)
    idx[0, 0], idx[0, 1] = 0, -3
    tower = mf.models.HashEmbeddingTower(997, d, num_hashes=num_hashes, seed=11, normalize=normalize, device=DEV)
    with torch.no_grad():
        tower.weight.copy_(table.to(DEV))
    assert torch.equal(tower.buckets(idx.to(DEV)).cpu(), oembed.hash_buckets(idx, num_hashes, 11, 997))   # bit-exact
    got = tower(idx.to(DEV)).detach().cpu()
    want = oembed.gather_hashed(table, idx, num_hashes, 11, normalize)
    if normalize:
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    else:
        assert torch.equal(got, want)


@pytest.mark.parametrize("opt", ["sgd", "adam"])
def test_hash_tower_training_step_matches_oracle(mf, opt):
    """Gradient of the normalised bloom sum reaches every bucket row of an id; collisions and duplicate
    ids are summed like any duplicate row."""
    g = torch.Generator().manual_seed(3)
    rows, d, n, nh = 61, 64, 300, 2                      # few buckets: plenty of collisions
    table0 = torch.randn(rows, d, generator=g)
    idx = torch.randint(0, 500, (n,), generator=g)
    tower = mf.models.HashEmbeddingTower(rows, d, num_hashes=nh, seed=5, device=DEV)
    with torch.no_grad():
        tower.weight.copy_(table0.to(DEV))
    optim = (mf.optim.SparseSGD(tower.parameters(), lr=0.1) if opt == "sgd" else mf.optim.RowAdam(tower.parameters(), lr=0.05))
    ref = table0.clone()
    m, v = torch.zeros_like(ref), torch.zeros_like(ref)
    b = oembed.hash_buckets(idx, nh, 5, rows)
    for step in (1, 2):
        gout = torch.randn(n, d, generator=g)
        tower(idx.to(DEV)).backward(gout.to(DEV))
        optim.step()
        optim.zero_grad()
        raw = ref[b[:, 0]] + ref[b[:, 1]]
        graw = oembed.normalize_backward(raw, gout)
        ids, grads = b.reshape(-1), graw.repeat_interleave(nh, dim=0)
        if opt == "sgd":
            oembed.sgd_update(ref, ids, grads, 0.1, 0.0)
        else:
            oembed.adam_update(ref, m, v, ids, grads, step=step, lr=0.05, weight_decay=0.01)
        torch.testing.assert_close(tower.weight.detach().cpu(), ref, rtol=5e-5, atol=5e-6)


# ------------------------------------------------------------- retrieval metrics (f-1) ---
@pytest.mark.parametrize("k", [1, 20, 64])
def test_retrieval_metrics_match_oracle(mf, k):
    g = torch.Generator().manual_seed(k)
    q, n_items = 97, 500
    topk = torch.stack([torch.randperm(n_items, generator=g)[:k] for _ in range(q)])
    topk[3, k // 2:] = -1                                     # a query with fewer than k results
    targets, off, ids, rel = [], [0], [], []
    for r in range(q):
        m = int(torch.randint(0, 40, (1,), generator=g)) if r != 5 else 0
        own = torch.randperm(n_items, generator=g)[:m].tolist()
        if m and r % 2 == 0:                                  # make sure some targets are retrieved
            own[: min(m, 3)] = topk[r, : min(m, 3)].tolist()
        own = [i for i in dict.fromkeys(own) if i >= 0]
        rat = torch.randint(0, 6, (len(own),), generator=g).tolist()   # rating 0: listed but not relevant
        targets.append(dict(zip(own, map(float, rat))))
        ids += own
        rel += rat
        off.append(len(ids))
    metric = mf.retrieval.RetrievalMetrics(top_k=k, prefix="val/")
    got = metric.update(topk.to(DEV), torch.tensor(off, device=DEV), torch.tensor(ids, dtype=torch.int64, device=DEV),
                        torch.tensor(rel, dtype=torch.float32, device=DEV)).cpu().numpy()
    want = oretr.retrieval_metrics(topk.numpy(), targets, k)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    agg = metric.compute()
    assert set(agg) == {"val/" + n for n in oretr.METRIC_NAMES}
    np.testing.assert_allclose(float(agg["val/RetrievalNormalizedDCG"]), want[:, 0].mean(), rtol=1e-5)


# ------------------------------------------------------------- batch producer (f-2) ---
def test_device_batch_producer_matches_oracle_and_feeds_the_module(mf):
    from oracle import data as odata

    g = torch.Generator().manual_seed(12)
    n_users, n_items, n_pairs = 200, 300, 1500
    pu = torch.randint(1, n_users, (n_pairs,), generator=g)
    pi = torch.randint(1, n_items, (n_pairs,), generator=g)
    pt = torch.randint(1, 6, (n_pairs,), generator=g).float()
    lists = [sorted(set(pi[pu == u].tolist())) for u in range(n_users)]       # a user's positives = its rated items
    off = torch.tensor([0] + list(np.cumsum([len(x) for x in lists])))
    items = torch.tensor([i for x in lists for i in x], dtype=torch.int64)
    longest = max(len(x) for x in lists)
    assert longest > 10
    with pytest.raises(ValueError, match="would drop positives"):        # the reference passes ALL targets (data/lightning.py:275-279)
        mf.data.DeviceInteractionSampler(pu, pi, pt, off, items, num_items=n_items, batch_size=64, pos_pad=10, device=DEV)
    with pytest.raises(ValueError, match="needs a row in pos_off"):
        mf.data.DeviceInteractionSampler(pu, pi, pt, off[:50], items, num_items=n_items, batch_size=64, device=DEV)
    full = mf.data.DeviceInteractionSampler(pu, pi, pt, off, items, num_items=n_items, batch_size=64, seed=77, device=DEV)
    assert full.pos_pad == 0                                                # default: CSR lists in place, no [B, P] tensor at all
    fb = full.batch(3)
    assert "pos_idx" not in fb["user"]
    uid, poff, pitems = fb["user"]["pos_csr"]
    assert uid is fb["user"]["idx"] and poff.data_ptr() == full.pos_off.data_ptr() and pitems.data_ptr() == full.pos_items.data_ptr()
    padded = mf.data.DeviceInteractionSampler(pu, pi, pt, off, items, num_items=n_items, batch_size=64, seed=77, device=DEV,
                                              pos_pad=longest)                # the reference's layout, sized so that nothing is dropped
    pb = padded.batch(3)
    assert torch.equal(pb["user"]["idx"], fb["user"]["idx"]) and torch.equal(pb["item"]["idx"], fb["item"]["idx"])
    for r, u in enumerate(pb["user"]["idx"].cpu().tolist()):
        assert [x for x in pb["user"]["pos_idx"][r].cpu().tolist() if x] == lists[u]
    # the same batch through both forms of the positives: identical losses, bit for bit (the masks are the same bits)
    mm = mf.lightning.MatrixFactorizationLitModule({"num_users": n_users, "num_items": n_items, "hidden_size": 32, "num_negatives": 0})
    mm.configure_model(device=DEV)
    la, lb = mm.compute_losses(fb), mm.compute_losses(pb)
    assert all(torch.equal(la[k], lb[k]) for k in la), {k: (float(la[k]), float(lb[k])) for k in la}
    # the per-rank stream of a user-sharded job: only pairs of users lo <= u < hi, each exactly once per epoch
    mine = mf.data.DeviceInteractionSampler(pu, pi, pt, off, items, num_items=n_items, batch_size=32, seed=5, device=DEV,
                                            user_range=(50, 120))
    n_mine = int(((pu >= 50) & (pu < 120)).sum())
    got_u = torch.cat([mine.batch(s_)["user"]["idx"] for s_ in range(mine.steps_per_epoch)])[:n_mine].cpu()
    assert int(got_u.min()) >= 50 and int(got_u.max()) < 120
    assert torch.equal(torch.sort(got_u).values, torch.sort(pu[(pu >= 50) & (pu < 120)]).values)
    sampler = mf.data.DeviceInteractionSampler(pu, pi, pt, off, items, num_items=n_items, batch_size=64, pos_pad=10, seed=77,
                                               device=DEV, truncate_positives=True)
    seen = []
    for step in (0, 5, 23, 24):                              # 23 -> 24 crosses the epoch boundary (1500 / 64 = 23.4)
        b = sampler.batch(step)
        u, it, t, pos = odata.sample_batch(pu.tolist(), pi.tolist(), pt.tolist(), off.tolist(), items.tolist(), n_items, 77,
                                           step * 64, 64, 10)
        assert torch.equal(b["user"]["idx"].cpu(), u)
        assert torch.equal(torch.cat([b["item"]["idx"], b["neg_item"]["idx"]]).cpu(), it)
        assert torch.equal(b["target"].cpu(), t) and torch.equal(b["user"]["pos_idx"].cpu(), pos)
        assert int(b["neg_item"]["idx"].min()) >= 1 and int(b["neg_item"]["idx"].max()) < n_items
        seen.append(b)
    assert torch.equal(sampler.batch(5)["user"]["idx"], seen[1]["user"]["idx"])          # a function of (seed, step)
    # one epoch visits every interaction exactly once
    epoch = torch.cat([sampler.batch(s)["item"]["idx"] for s in range(24)])[:n_pairs].cpu()
    assert torch.equal(torch.sort(epoch).values, torch.sort(pi).values)
    # and the batch goes straight into the training step
    m = mf.lightning.MatrixFactorizationLitModule({"num_users": n_users, "num_items": n_items, "hidden_size": 32})
    m.configure_model(device=DEV)
    loss = m.training_step(seen[0], 0)
    assert torch.isfinite(loss)


@pytest.mark.parametrize("n,nk", [(1, 1), (1000, 2), (24576, 8), (32768, 64), (5000, 3)])
def test_group_keys_is_a_stable_counting_sort(mf, n, nk):
    """mf_group_keys (the exchange plans' grouping by owner rank) against torch: stable permutation, sorted keys, group bounds."""
    g = torch.Generator().manual_seed(n + nk)
    keys = torch.randint(0, nk, (n,), generator=g)
    if n > 100:
        keys[: n // 3] = nk - 1                                      # a long run: whole waves of the first chunks hold one key
    got = mf.distributed.HipOps(mf).group_by_key(keys.to(DEV), nk)
    assert got is not None
    perm, sk, bounds = (t.cpu() for t in got)
    want = torch.argsort(keys, stable=True)
    assert torch.equal(perm, want)
    assert torch.equal(sk, keys[want])
    assert torch.equal(bounds, torch.cat([torch.zeros(1, dtype=torch.int64), torch.bincount(keys, minlength=nk).cumsum(0)]))
    assert mf.distributed.HipOps(mf).group_by_key(torch.zeros(40000, dtype=torch.int64, device=DEV), 8) is None      # beyond the limits


@pytest.mark.parametrize("adam", [True, False])
def test_update_pair_equals_two_updates(mf, adam):
    """``mf_update_pair`` -- both tables of a step in ONE launch -- against the two ``mf_update_adam`` / ``mf_update_sgd`` calls it
    replaces: torch.equal on tables and moments (Zipf ids with a run of hundreds of duplicates, out-of-range ids, two steps)."""
    lib = mf._lib.lib()
    g = torch.Generator().manual_seed(3)
    d, rows_a, rows_b, n_a, n_b = 128, 5000, 700, 4096, 8192
    w = 1.0 / torch.arange(1, rows_b + 1, dtype=torch.float64)
    ia = torch.randint(-2, rows_a + 2, (n_a,), generator=g).to(DEV)
    ib = torch.multinomial(w, n_b, replacement=True, generator=g).to(DEV)
    ga, gb = torch.randn(n_a, d, generator=g).to(DEV), torch.randn(n_b, d, generator=g).to(DEV)

    def tables():
        t = torch.Generator().manual_seed(9)
        ta, tb = torch.randn(rows_a, d, generator=t).to(DEV), torch.randn(rows_b, d, generator=t).to(DEV)
        return ta, tb, [torch.zeros_like(ta), torch.zeros_like(ta)], [torch.zeros_like(tb), torch.zeros_like(tb)]

    (ta1, tb1, sa1, sb1), (ta2, tb2, sa2, sb2) = tables(), tables()
    wa = mf._lib.workspace(lib.mf_update_ws_bytes(n_a, d), DEV)
    wb = mf._lib.workspace(lib.mf_update_ws_bytes(n_b, d), DEV)
    for step in (1, 2):
        for t, s, i, gr, ws in ((ta1, sa1, ia, ga, wa), (tb1, sb1, ib, gb, wb)):
            if adam:
                mf._lib.check(lib.mf_update_adam(t.data_ptr(), s[0].data_ptr(), s[1].data_ptr(), t.shape[0], d, i.data_ptr(), i.numel(), gr.data_ptr(), 1,
                                                 step, None, 0.05, 0.9, 0.999, 1e-8, 0.01, ws.data_ptr(), ws.numel(), None))
            else:
                mf._lib.check(lib.mf_update_sgd(t.data_ptr(), t.shape[0], d, i.data_ptr(), i.numel(), gr.data_ptr(), 1, 0.05, 0.01, ws.data_ptr(),
                                                ws.numel(), None))
        mf._lib.check(lib.mf_update_pair(int(adam), d, ta2.data_ptr(), sa2[0].data_ptr() if adam else None, sa2[1].data_ptr() if adam else None, rows_a,
                                         ia.data_ptr(), n_a, ga.data_ptr(), 1, wa.data_ptr(), wa.numel(), tb2.data_ptr(),
                                         sb2[0].data_ptr() if adam else None, sb2[1].data_ptr() if adam else None, rows_b, ib.data_ptr(), n_b,
                                         gb.data_ptr(), 1, wb.data_ptr(), wb.numel(), step, None, 0.05, 0.9, 0.999, 1e-8, 0.01, None))
        assert torch.equal(ta1, ta2) and torch.equal(tb1, tb2), step
        if adam:
            assert all(torch.equal(x, y) for x, y in zip(sa1 + sb1, sa2 + sb2)), step


def _mined_mask(mf, t, k, sigma, logq=None):
    kw = dict(item_idx=t["item_idx"].to(DEV), pos_idx=t["pos_idx"].to(DEV), num_negatives=k, sigma=sigma)
    return mf.losses.negative_mask(t["u"].to(DEV), t["v"].to(DEV), t["target"].to(DEV), **kw).cpu()


@pytest.mark.parametrize("tied", [False, True, "ties-under-distinct-ids"], ids=["distinct", "zipf-copies", "gives-up"])
@pytest.mark.parametrize("cfg", [(512, 2048, 64, 4, 1.0), (300, 2500, 128, 4, 1.0), (1024, 4096, 128, 32, 30.0), (700, 3000, 64, 9, 1000.0)],
                         ids=lambda c: "x".join(map(str, c)))
def test_mined_masks_through_the_bf16_prefilter_are_bit_exact(mf, cfg, tied):
    """The split-bf16 candidate search (csrc/mf_mine_bf.h) against the oracle's semi-hard mining AND against the fp32
    streaming selection it replaces at these shapes: all three masks are the same bits -- zero-target users, duplicate
    item rows (exact ties), accidental hits and a ragged last tile included.  `tied`: the item rows are looked up by id in a
    table, as the module does, with a popularity skew -- hundreds of bit-identical columns, which the prefilter scans once
    and expands behind the winners; every fifth user's positive is missing from its own list (its copies are then valid
    negatives at Dm = 0 exactly)."""
    b, n, d, k, sigma = cfg
    t = _random_case(b, n, d, 6, seed=sum(map(int, cfg[:4])), n_items=n // 3)
    if tied == "ties-under-distinct-ids":
        # a thousand bit-identical rows under DIFFERENT ids (nothing tells the prefilter they are copies): every list and spill
        # list overflows, the batch is handed to the fp32 search on the device -- same bits again
        t["v"][n // 4:n // 4 + 1000] = t["v"][3]
    elif tied:
        g = torch.Generator().manual_seed(7)
        ids = torch.randint(1, n // 8, (n,), generator=g)
        ids[40:40 + n // 4] = 3                              # one item a quarter of the batch
        ids[n - 300:n - 100] = 5
        table = _unit(n // 8 + 1, d, g)
        t["item_idx"] = ids
        t["v"] = table[ids].clone()
        t["pos_idx"] = torch.randint(0, n // 8, t["pos_idx"].shape, generator=g)
        t["pos_idx"][:, 0] = ids[:b]
        t["pos_idx"][::5, 0] = 0                             # (0 = padding: these users do not list their own positive)
    else:
        t["v"][n // 2:n // 2 + 40] = t["v"][:40]             # duplicate rows under different ids: equal scores, the column decides
    t["u"][5] = t["u"][4]
    lg = torch.from_numpy(chain.logits(t["u"].numpy(), t["v"].numpy(), t["target"].numpy(), sigma))
    want = ol.semi_hard_mining(lg, ol.negative_masks(t["item_idx"], t["pos_idx"], b), k)
    lib = mf._lib.lib()
    try:
        lib.mf_set_mining_prefilter(0)
        plain = _mined_mask(mf, t, k, sigma)
        lib.mf_set_mining_prefilter(2)                  # (2: wherever it can serve -- these shapes are below the default's B >= 4096)
        got = _mined_mask(mf, t, k, sigma)
    finally:
        lib.mf_set_mining_prefilter(1)
    assert torch.equal(plain, want), int((plain != want).sum())
    assert torch.equal(got, want), (int((got != want).sum()), (got != want).any(1).nonzero().flatten()[:8].tolist())
    if tied is True and k <= 9:
        # a logQ tensor PER COLUMN: copies of an item may carry different values, and are then no copies to the search
        # (the randomised stress found the first version treating them as such)
        g = torch.Generator().manual_seed(11)
        logq = torch.log(torch.rand(n, generator=g) * 0.9 + 0.05)
        lgq = chain.logits(t["u"].numpy(), t["v"].numpy(), t["target"].numpy(), sigma, logq.numpy())
        ref = float(ol.loss("PairwiseHingeLoss", t["u"], t["v"], t["target"], item_idx=t["item_idx"], pos_idx=t["pos_idx"], num_negatives=k,
                            sigma=sigma, margin=0.5, logq=logq, mining_logits=lgq))
        try:
            lib.mf_set_mining_prefilter(2)
            val = _run_gpu(mf, "PairwiseHingeLoss", t, k, sigma, 0.5, logq)[0]
        finally:
            lib.mf_set_mining_prefilter(1)
        assert abs(val - ref) <= gu.loss_tolerance(ref, sigma, t["target"].numpy()), (val, ref)
