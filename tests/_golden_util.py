"""Shared by the CPU (oracle) and GPU tests of the ``wide_*`` fixtures (sigma up to 1000, negative margin, k = 32:
the ends of the range the reference's tuners visit, xfmr_rec/ray.py:147-149).

At sigma = 1000 two implementations of the same formulas agree on a logit to ~1e-6 * sigma, which is enough to move a
hinge element across its kink or to swap two near-tied columns at the mining cut -- in the reference as much as here
(its own cdist rounding decides).  Such elements are found from the reference's stored logits and masks and left out of
the GRADIENT comparison (the loss value is continuous across both events and is always compared); the tests assert that
only a small fraction is left out."""
from __future__ import annotations

import numpy as np

HINGE_KINDS = {1: "contrastive", 2: "contrastive", 5: "pairwise"}       # index into KINDS -> where the relu sits


def unpack_mask(z, k, smi, b, n):
    return np.unpackbits(z[f"mask_{k}_{smi}"])[: b * n].reshape(b, n).astype(bool)


def assert_masks_equal_up_to_ties(z, got, k, smi, what, tol=0.0):
    """identical sets, or differing only inside a group the reference's own sort key ties (duplicate items give equal
    logits; torch.topk leaves the choice among them unspecified) -- ``tol`` > 0 widens a tie to near-ties (another
    implementation's logits, differing by rounding)."""
    b, n = z["u"].shape[0], z["v"].shape[0]
    want = unpack_mask(z, k, smi, b, n)
    lg = z[f"logits_{smi}"].astype(np.float64)
    dm = lg - np.diagonal(lg)[:, None]
    ref_key = np.where(dm < 0, dm - dm.min(-1, keepdims=True), -dm)
    for i in np.nonzero((got != want).any(1))[0]:
        assert got[i].sum() == want[i].sum(), (what, i)
        a, c = np.sort(ref_key[i][got[i]]), np.sort(ref_key[i][want[i]])
        assert np.abs(a - c).max() <= tol, (what, i, np.abs(a - c).max())


def undecided(z, ki, k, smi, sigma, margin, got_mask=None, tol=1e-5):
    """(rows of du, rows of dv) the comparison skips for loss kind ``ki``; ``got_mask``: the tested implementation's
    post-mining mask (None: same as the reference's)."""
    target = z["target"]
    b, n = z["u"].shape[0], z["v"].shape[0]
    lg = z[f"logits_{smi}"].astype(np.float64)
    want = unpack_mask(z, k, smi, b, n)
    got = want if got_mask is None else np.asarray(got_mask, dtype=bool)
    rows, cols = set(), set()
    for i in np.nonzero((got != want).any(1))[0]:          # another set of mined negatives: the whole row's share moves
        if target[i] == 0:                                 # (sign 0: every logit of the row is 0 and ties; weight 0: no gradient)
            continue
        rows.add(int(i))
        cols.update(np.nonzero(got[i] | want[i])[0].tolist() + [int(i)])
    if ki in HINGE_KINDS:
        sgn = np.sign(target).astype(np.float64)[:, None]
        x = lg + sgn * margin if HINGE_KINDS[ki] == "contrastive" else lg - np.diagonal(lg)[:, None] + margin
        for i, j in zip(*np.nonzero((np.abs(x) < tol * sigma) & (want | got) & (target != 0)[:, None])):
            rows.add(int(i))
            cols.update((int(i), int(j)))
    return rows, cols


def keep(total, skipped, stride=1):
    """indices (into the ``[::stride]`` sample the fixture stores) of the rows to compare."""
    sample = np.arange(0, total, stride)
    ok = np.array([r not in skipped for r in sample], dtype=bool)
    return np.nonzero(ok)[0]


def assert_grads_close(got, want, sigma, what, base=1e-4, per_sigma=5e-6, floor=1e-5):
    """Row-wise relative comparison.  A gradient row is (coefficients that are exp / sigmoid of logit differences) x
    (embedding rows); two implementations agree on a logit to a few ulp of the squared distance times sigma, so the
    coefficients agree to a few 1e-6 * sigma RELATIVE (exp'(x) = exp(x)): the tolerance is ``base + per_sigma * sigma`` of
    each row's largest entry -- 1e-4 at sigma = 1 (the north-star's bar), 5e-3 at sigma = 1000 -- plus an absolute
    ``floor * sigma`` (1 - softmax_ii formed in fp32: an absolute ~1e-7 x |target| (<= 5) x sigma x |u - v| on rows whose
    negatives carry almost no probability; the older fixtures' tests allow 2e-5 * sigma)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if want.size == 0:
        return
    scale = np.abs(want).max(axis=1, keepdims=True)
    tol = base + per_sigma * sigma
    err = np.abs(got - want)
    bad = err > tol * (np.abs(want) + scale) + floor * sigma
    assert not bad.any(), (what, int(bad.sum()), float((err / (scale + 1e-30)).max()), tol)


def loss_tolerance(want, sigma, target, rel=1e-5, per_logit=2e-6):
    """|loss - reference| allowed: ``rel`` of the value plus what a logit error of ``per_logit * sigma`` (a few ulp of the
    squared distance, times sigma) moves a weighted sum of per-row losses by -- every loss here is 1-Lipschitz in a logit.
    Always inside the north-star's 1e-4 * sigma * max(1, |loss|) for the fixtures' weights."""
    return rel * abs(float(want)) + per_logit * sigma * float(np.abs(np.asarray(target, dtype=np.float64)).sum())
