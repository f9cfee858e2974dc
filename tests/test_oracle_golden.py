"""CPU: pin oracle/losses.py to the reference's own outputs (tests/golden/*.npz,
made by tests/golden/make_golden.py from /root/reference/xfmr_rec/losses.py)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import losses as ol
from oracle import embed as oembed
from tests import _golden_util as gu
from tests.conftest import GOLDEN, golden_files

SIGMA_MARGIN = ((1.0, 1.0), (2.0, 0.5), (1.0, 0.0))
FILES = golden_files()


def _load(path):
    z = np.load(path)
    t = {k: torch.from_numpy(z[k]) for k in ("u", "v", "target", "item_idx", "pos_idx")}
    return z, t


def test_golden_present():
    assert len(FILES) == 6


@pytest.mark.parametrize("path", FILES, ids=lambda p: p.stem)
def test_masks_match_reference(path):
    """negative_masks + semi_hard_mining: identical sets, or differing only inside a
    group the reference's own sort key ties (torch.topk leaves that unspecified)."""
    z, t = _load(path)
    B, N = t["u"].shape[0], t["v"].shape[0]
    for smi, (sigma, _) in enumerate(SIGMA_MARGIN):
        lg = torch.from_numpy(z[f"logits_{smi}"])
        for k in (0, 4, N):
            want = np.unpackbits(z[f"mask_{k}_{smi}"])[: B * N].reshape(B, N).astype(bool)
            got = ol.semi_hard_mining(lg, ol.negative_masks(t["item_idx"], t["pos_idx"], B), k).numpy()
            if (got == want).all():
                continue
            dm = lg - lg.diagonal()[:, None]
            ref_key = torch.where(dm < 0, dm - dm.min(-1, keepdim=True)[0], -dm).numpy()
            for i in np.nonzero((got != want).any(1))[0]:
                assert got[i].sum() == want[i].sum()
                assert np.array_equal(np.sort(ref_key[i][got[i]]), np.sort(ref_key[i][want[i]])), (
                    path.stem, smi, k, i)


@pytest.mark.parametrize("path", FILES, ids=lambda p: p.stem)
def test_hard_mining_matches_reference(path):
    """hard_mining (losses.py:112-132) on the reference's own logits and masks: identical sets, or differing only
    among equal logits at the cut (torch.topk leaves ties unspecified)."""
    z, t = _load(path)
    B, N = t["u"].shape[0], t["v"].shape[0]
    for smi in range(len(SIGMA_MARGIN)):
        lg = torch.from_numpy(z[f"logits_{smi}"])
        want = np.unpackbits(z[f"hard_4_{smi}"])[: B * N].reshape(B, N).astype(bool)
        got = ol.hard_mining(lg, ol.negative_masks(t["item_idx"], t["pos_idx"], B), 4).numpy()
        for i in np.nonzero((got != want).any(1))[0]:
            assert got[i].sum() == want[i].sum()
            assert np.array_equal(np.sort(lg[i].numpy()[got[i]]), np.sort(lg[i].numpy()[want[i]])), (path.stem, smi, i)


@pytest.mark.parametrize("path", FILES, ids=lambda p: p.stem)
def test_losses_and_grads_match_reference(path):
    z, t = _load(path)
    N = t["v"].shape[0]
    gs = int(z["gstride"])
    for ki, kind in enumerate(ol.KINDS):
        for k in (0, 4, N):
            for smi, (sigma, margin) in enumerate(SIGMA_MARGIN):
                tag = f"{ki}_{k}_{smi}"
                u = t["u"].clone().requires_grad_()
                v = t["v"].clone().requires_grad_()
                val = ol.loss(kind, u, v, t["target"], item_idx=t["item_idx"], pos_idx=t["pos_idx"],
                              num_negatives=k, sigma=sigma, margin=margin,
                              mining_logits=z[f"logits_{smi}"])
                want = float(z[f"loss_{tag}"])
                if not np.isfinite(want):          # MINE with no valid negative: -inf (losses.py:242-244)
                    assert float(val) == want, (tag, float(val.detach()), want)
                    continue
                assert abs(float(val.detach()) - want) <= 1e-5 * max(1.0, abs(want)), (tag, float(val.detach()), want)
                val.backward()
                np.testing.assert_allclose(u.grad.numpy()[::gs], z[f"du_{tag}"], rtol=1e-4, atol=2e-6, err_msg=tag)
                np.testing.assert_allclose(v.grad.numpy()[::gs], z[f"dv_{tag}"], rtol=1e-4, atol=2e-6, err_msg=tag)


# ------------------------------------------------ the ends of the reference's hyper-parameter range (ray.py:147-149) ---
WIDE = sorted(GOLDEN.glob("wide_*.npz"))


def test_wide_and_step_fixtures_present():
    assert len(WIDE) == 2 and (GOLDEN / "step_B32_N64_d32_P16.npz").exists()


@pytest.mark.parametrize("path", WIDE, ids=lambda p: p.stem)
def test_wide_range_losses_and_grads_match_reference(path):
    """sigma = 30 / 1000, margin = -0.5, num_negatives = 1 / 4 / 32: the oracle against the reference's own values.  The
    mining decisions use the reference's logits (as above); gradient rows a kink decides are left out (tests/_golden_util.py)."""
    z, t = _load(path)
    b, n = t["u"].shape[0], t["v"].shape[0]
    gs = int(z["gstride"])
    skipped = total = 0
    for smi, (sigma, margin) in enumerate(z["sigma_margin"].tolist()):
        for k in z["ks"].tolist():
            got_mask = ol.semi_hard_mining(torch.from_numpy(z[f"logits_{smi}"]), ol.negative_masks(t["item_idx"], t["pos_idx"], b), k)
            got_mask = got_mask.numpy()
            gu.assert_masks_equal_up_to_ties(z, got_mask, k, smi, (path.stem, smi, k))
            for ki, kind in enumerate(ol.KINDS):
                tag = f"{ki}_{k}_{smi}"
                u = t["u"].clone().requires_grad_()
                v = t["v"].clone().requires_grad_()
                val = ol.loss(kind, u, v, t["target"], item_idx=t["item_idx"], pos_idx=t["pos_idx"], num_negatives=k,
                              sigma=sigma, margin=margin, mining_logits=z[f"logits_{smi}"])
                want = float(z[f"loss_{tag}"])
                assert abs(float(val.detach()) - want) <= gu.loss_tolerance(want, sigma, z["target"]), (tag, float(val.detach()), want)
                val.backward()
                rows, cols = gu.undecided(z, ki, k, smi, sigma, margin, got_mask)
                ru, rv = gu.keep(b, rows, gs), gu.keep(n, cols, gs)
                skipped += (len(range(0, b, gs)) - len(ru)) + (len(range(0, n, gs)) - len(rv))
                total += len(range(0, b, gs)) + len(range(0, n, gs))
                gu.assert_grads_close(u.grad.numpy()[::gs][ru], z[f"du_{tag}"][ru], sigma, ("du", tag))
                gu.assert_grads_close(v.grad.numpy()[::gs][rv], z[f"dv_{tag}"][rv], sigma, ("dv", tag))
    assert skipped <= 0.05 * total, (skipped, total)


def _step_fixture():
    z = np.load(GOLDEN / "step_B32_N64_d32_P16.npz")
    t = {k: torch.from_numpy(z[k]) for k in ("U", "V", "user", "item", "target", "pos_idx")}
    return z, t


def test_table_consistent_step_matches_reference():
    """The whole default step (xfmr_rec/params.py:18, lightning.py:33,38-39,189-192,238-239) on the CPU oracle -- gather +
    L2-normalise, the seven losses, backward through the normalisation, SGD / AdamW on the touched rows -- against what the
    reference's losses + torch.autograd + torch.optim produce from the same tables and ids (duplicate ids share rows)."""
    z, t = _step_fixture()
    k = int(z["num_negatives"])
    u0, v0 = oembed.gather(t["U"], t["user"]), oembed.gather(t["V"], t["item"])
    for ki, kind in enumerate(ol.KINDS):
        u, v = u0.clone().requires_grad_(), v0.clone().requires_grad_()
        val = ol.loss(kind, u, v, t["target"], item_idx=t["item"], pos_idx=t["pos_idx"], num_negatives=k)
        want = float(z[f"loss_{ki}"])
        assert abs(float(val.detach()) - want) <= 1e-5 * max(1.0, abs(want)), (kind, float(val.detach()), want)
        val.backward()
        np.testing.assert_allclose(u.grad.numpy(), z[f"du_{ki}"], rtol=1e-4, atol=2e-6, err_msg=kind)
        # duplicate item ids share a table row here, so their logits tie exactly and the two minings may keep different
        # copies of one item (torch.topk leaves that open): per ITEM the gradient is the same, per column it need not be
        got_dv, want_dv = oembed.coalesce(t["item"], v.grad)[1], oembed.coalesce(t["item"], torch.from_numpy(z[f"dv_{ki}"]))[1]
        np.testing.assert_allclose(got_dv.numpy(), want_dv.numpy(), rtol=1e-4, atol=5e-6, err_msg=kind)
        gu_raw = oembed.normalize_backward(t["U"][t["user"]], torch.from_numpy(z[f"du_{ki}"]))
        gv_raw = oembed.normalize_backward(t["V"][t["item"]], torch.from_numpy(z[f"dv_{ki}"]))
        U, V = t["U"].clone(), t["V"].clone()
        oembed.sgd_update(U, t["user"], gu_raw, float(z["lr_sgd"]))
        oembed.sgd_update(V, t["item"], gv_raw, float(z["lr_sgd"]))
        np.testing.assert_allclose(U.numpy(), z[f"U_sgd_{ki}"], rtol=0, atol=2e-6, err_msg=kind)
        np.testing.assert_allclose(V.numpy(), z[f"V_sgd_{ki}"], rtol=0, atol=2e-6, err_msg=kind)
        for tab, ids, g, name in ((t["U"], t["user"], gu_raw, "U"), (t["V"], t["item"], gv_raw, "V")):
            w, m, s = tab.clone(), torch.zeros_like(tab), torch.zeros_like(tab)
            oembed.adam_update(w, m, s, ids, g, step=1, lr=float(z["lr_adam"]))
            dense = torch.from_numpy(z[f"d{name}_{ki}"])
            firm = dense.abs() > 1e-4            # at step 1 the update is lr * g / (|g| + eps): tiny g are decided by rounding
            np.testing.assert_allclose(w.numpy()[firm], z[f"{name}_adam_{ki}"][firm], rtol=0, atol=5e-6, err_msg=kind)
