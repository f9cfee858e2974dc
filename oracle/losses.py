"""CPU restatement of ``xfmr_rec/losses.py`` (TEST INFRASTRUCTURE; parity PINNED by
tests/golden/*.npz, see oracle/__init__.py).

Written from the formulas, not from the reference's op
sequence: scores come from ``u @ v.T`` and row norms, the positive mask is built by
sorting the batch's item ids and range-searching each user's positives
(O((N + B*P) log N) instead of the reference's B x N x P boolean temp,
losses.py:108), and mining uses the explicit total order of
include/mf_numerics.h.  Everything is differentiable torch, so ``autograd`` of this
file is the independent check of the hand-derived HIP backward.
"""
from __future__ import annotations

import torch

KINDS = (
    "AlignmentLoss",                               # losses.py:249-259
    "ContrastiveLoss",                             # losses.py:262-274
    "AlignmentContrastiveLoss",                    # losses.py:277-291
    "InfomationNoiseContrastiveEstimationLoss",    # losses.py:294-306 (sic)
    "MutualInformationNeuralEstimationLoss",       # losses.py:309-321
    "PairwiseHingeLoss",                           # losses.py:357-359
    "PairwiseLogisticLoss",                        # losses.py:352-354
)


def half_sqdist(u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """0.5 * ||u_i - v_j||^2  (losses.py:9-12, cdist(u, v)**2 / 2)."""
    nu = (u * u).sum(-1)
    nv = (v * v).sum(-1)
    sq = (nu[:, None] + nv[None, :] - 2.0 * (u @ v.T)).clamp_min(0.0)
    return 0.5 * sq


def logits_fn(u, v, target, sigma: float, logq: torch.Tensor | None = None) -> torch.Tensor:
    """L_ij = -D_ij * sign(t_i) * sigma (losses.py:181-183) [- log q_j: our spec]."""
    lg = -half_sqdist(u, v) * torch.sign(target).to(u.dtype)[:, None] * sigma
    if logq is not None:
        lg = lg - logq[None, :].to(u.dtype)
    return lg


@torch.no_grad()
def negative_masks(item_idx: torch.Tensor, pos_idx: torch.Tensor | None, batch: int) -> torch.Tensor:
    """M[i, j] = not(item_idx[i] == item_idx[j] or item_idx[j] in pos_idx[i, :])
    (losses.py:92-110).  No special case for the 0 padding: like the reference, a
    padded 0 matches a column whose item id is 0."""
    n = item_idx.numel()
    order = torch.argsort(item_idx, stable=True)
    skeys = item_idx[order]
    keys = item_idx[:batch, None]
    if pos_idx is not None:
        keys = torch.cat([keys, pos_idx.to(item_idx.dtype)], dim=1)
    lo = torch.searchsorted(skeys, keys.contiguous(), right=False)
    hi = torch.searchsorted(skeys, keys.contiguous(), right=True)
    cnt = (hi - lo).reshape(-1)
    rows = torch.arange(batch).repeat_interleave(keys.shape[1]).repeat_interleave(cnt)
    start = lo.reshape(-1).repeat_interleave(cnt)
    within = torch.arange(int(cnt.sum())) - (cnt.cumsum(0) - cnt).repeat_interleave(cnt)
    cols = order[start + within]
    hit = torch.zeros(batch, n, dtype=torch.bool)
    hit[rows, cols] = True
    return ~hit


@torch.no_grad()
def semi_hard_mining(lg: torch.Tensor, neg: torch.Tensor, k: int) -> torch.Tensor:
    """losses.py:134-162.  Keeps, per row, the min(k, #valid) valid negatives that
    come first in: semi-hard (L_ij < L_ii) by descending L, then hard by ascending
    L, lowest column on exact ties (the refinement documented in mf_numerics.h)."""
    n = lg.shape[1]
    if k <= 0 or k >= n:
        return neg
    dm = lg - lg.diagonal()[:, None]
    cls = torch.where(dm < 0, 2, 1) * neg.to(torch.int64)      # 2 semi, 1 hard, 0 masked
    val = torch.where(dm < 0, dm, -dm)
    o1 = torch.argsort(val, dim=1, descending=True, stable=True)
    o2 = torch.argsort(cls.gather(1, o1), dim=1, descending=True, stable=True)
    top = o1.gather(1, o2)[:, :k]
    sel = torch.zeros_like(neg)
    sel.scatter_(1, top, True)
    return neg & sel


@torch.no_grad()
def hard_mining(lg: torch.Tensor, neg: torch.Tensor, k: int) -> torch.Tensor:
    """losses.py:112-132 (defined upstream, never called): keeps, per row, the min(k, #valid) valid negatives with
    the highest logits, lowest column on exact ties (torch.topk leaves ties unspecified)."""
    n = lg.shape[1]
    if k <= 0 or k >= n:
        return neg
    key = torch.where(neg, lg, torch.full_like(lg, float("-inf")))
    top = torch.argsort(key, dim=1, descending=True, stable=True)[:, :k]
    sel = torch.zeros_like(neg)
    sel.scatter_(1, top, True)
    return neg & sel


def _wmean(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """losses.py:15-23."""
    wf = w.to(x.dtype)
    return (x * wf / (wf.sum(-1, keepdim=True) + 1e-10)).sum(-1)


def _masked_lse(lg: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    return torch.logsumexp(torch.where(m, lg, torch.full_like(lg, float("-inf"))), dim=-1)


def loss(kind: str, u, v, target, *, item_idx, pos_idx, num_negatives: int = 0,
         sigma: float = 1.0, margin: float = 1.0, logq=None, mining_logits=None,
         return_mask: bool = False):
    """One of the seven losses; ``mining_logits`` optionally supplies exact chain
    logits (oracle.chain.logits) for the no-grad mask/mining decisions."""
    assert kind in KINDS, kind
    b = u.shape[0]
    tf = target.to(u.dtype)
    sgn, w = torch.sign(tf), tf.abs()
    align = (0.5 * ((u - v[:b]) ** 2).sum(-1) * tf * sigma).sum()          # losses.py:164-170
    if kind == "AlignmentLoss":
        return (align, None) if return_mask else align
    lg = logits_fn(u, v, tf, sigma, logq)
    dec = lg.detach() if mining_logits is None else torch.as_tensor(mining_logits)
    m = semi_hard_mining(dec, negative_masks(item_idx, pos_idx, b), num_negatives)
    diag = lg.diagonal()
    if kind in ("ContrastiveLoss", "AlignmentContrastiveLoss"):            # losses.py:172-193
        per = _wmean(torch.relu(lg + sgn[:, None] * margin), m)
        out = (per * w).sum()
        if kind == "AlignmentContrastiveLoss":
            out = out + align
    elif kind == "InfomationNoiseContrastiveEstimationLoss":               # losses.py:195-223
        eye = torch.eye(b, lg.shape[1], dtype=torch.bool)
        out = ((_masked_lse(lg, m | eye) - diag) * w).sum()
    elif kind == "MutualInformationNeuralEstimationLoss":                  # losses.py:225-246
        out = ((_masked_lse(lg, m) - diag) * w).sum()
    else:                                                                  # losses.py:324-346
        x = lg - diag[:, None] + margin
        phi = torch.relu(x) if kind == "PairwiseHingeLoss" else torch.nn.functional.softplus(x)
        out = (_wmean(phi, m) * w).sum()
    return (out, m) if return_mask else out


def all_losses(u, v, target, *, item_idx, pos_idx, num_negatives=0, sigma=1.0, margin=1.0,
               logq=None) -> dict[str, torch.Tensor]:
    """What compute_losses evaluates every step (xfmr_rec/lightning.py:137-146)."""
    return {
        k: loss(k, u, v, target, item_idx=item_idx, pos_idx=pos_idx,
                num_negatives=num_negatives, sigma=sigma, margin=margin, logq=logq)
        for k in KINDS
    }
