"""CPU: pin oracle/losses.py to the reference's own outputs (tests/golden/*.npz,
made by tests/golden/make_golden.py from /root/reference/xfmr_rec/losses.py)."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import losses as ol
from tests.conftest import golden_files

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
