#!/usr/bin/env python3
"""Generate the golden vectors that pin oracle/losses.py to the reference.

Runs ONLY in the build container (it imports the reference's own
``/root/reference/xfmr_rec/losses.py`` by path, read-only, no bytecode written).
The outputs -- inputs and the reference's outputs, i.e. data -- are committed as
``tests/golden/losses_*.npz``; nothing of the reference travels to the GPU box.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

For every shape x the 7 loss classes x num_negatives in {0, 4, N} x
(sigma, margin) in {(1,1), (2,0.5), (1,0)} it stores the scalar loss, the
post-mining boolean mask (packed bits) and d loss/d user_embed, d loss/d item_embed
(every ``gstride``-th row for the larger shapes, to keep the fixtures small), and -- per (sigma, margin) -- the
mask ``hard_mining`` (losses.py:112-132, k = 4) leaves of the same negative mask.
"""
from __future__ import annotations

import importlib.util
import pathlib
import sys

import numpy as np
import torch

HERE = pathlib.Path(__file__).resolve().parent
REF = pathlib.Path("/root/reference/xfmr_rec/losses.py")

KINDS = (
    "AlignmentLoss", "ContrastiveLoss", "AlignmentContrastiveLoss",
    "InfomationNoiseContrastiveEstimationLoss", "MutualInformationNeuralEstimationLoss",
    "PairwiseHingeLoss", "PairwiseLogisticLoss",
)
SIGMA_MARGIN = ((1.0, 1.0), (2.0, 0.5), (1.0, 0.0))
# (B, N, d, P, grad row stride, edge-case flag)
SHAPES = (
    (4, 8, 16, 3, 1, False),
    (4, 8, 16, 8, 1, True),       # row 3: no valid negative (MINE = -inf); row 2: u == v
    (32, 64, 32, 16, 2, False),
    (48, 96, 64, 33, 6, False),
    (64, 128, 128, 64, 16, False),
    (256, 512, 64, 24, 32, False),     # 2 x 4 sweep tiles of 128 x 128: the multi-block paths against the reference itself
)


def load_reference():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("_ref_losses", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_inputs(B, N, d, P, edge, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.nn.functional.normalize(torch.randn(B, d, generator=g), dim=-1)
    v = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=-1)
    target = torch.randint(1, 6, (B,), generator=g)
    if B >= 4:
        target[1] = 0          # zero weight, sign 0
        target[2] = -3         # negative feedback flips the logit sign
    n_items = max(N // 2, 4)   # small id range => duplicates inside the batch
    item_idx = torch.randint(1, n_items + 1, (N,), generator=g)
    item_idx[B] = item_idx[0]  # a sampled negative that IS row 0's positive
    pos_idx = torch.zeros(B, P, dtype=torch.long)
    for i in range(B):
        n_pos = int(torch.randint(1, P + 1, (1,), generator=g))
        extra = torch.randint(1, n_items + 1, (n_pos,), generator=g)
        extra[0] = item_idx[i]
        pos_idx[i, :n_pos] = extra
    if edge:
        item_idx = torch.arange(1, N + 1)
        item_idx[B] = item_idx[0]
        pos_idx.zero_()
        for i in range(B):
            pos_idx[i, 0] = item_idx[i]
        pos_idx[3, :] = torch.arange(1, P + 1)      # every column is a positive of row 3
        target[1], target[2] = 2, 4
        v[2] = u[2]                                  # zero distance on the diagonal
    return u, v, target, item_idx, pos_idx


def main() -> None:
    ref = load_reference()
    torch.set_num_threads(1)
    for si, (B, N, d, P, gstride, edge) in enumerate(SHAPES):
        u0, v0, target, item_idx, pos_idx = make_inputs(B, N, d, P, edge, seed=1234 + si)
        out = {
            "u": u0.numpy(), "v": v0.numpy(), "target": target.numpy(),
            "item_idx": item_idx.numpy(), "pos_idx": pos_idx.numpy(),
            "gstride": np.int64(gstride),
        }
        for ki, kind in enumerate(KINDS):
            for k in (0, 4, N):
                for smi, (sigma, margin) in enumerate(SIGMA_MARGIN):
                    fn = getattr(ref, kind)(num_negatives=k, sigma=sigma, margin=margin)
                    u = u0.clone().requires_grad_()
                    v = v0.clone().requires_grad_()
                    val = fn(u, v, target, item_idx=item_idx, pos_idx=pos_idx)
                    val.backward()
                    with torch.no_grad():
                        lg = -ref.squared_distance(u, v) * target.sign()[:, None] * sigma
                        mask = fn.negative_masks(lg, item_idx=item_idx, pos_idx=pos_idx)
                        mask = fn.semi_hard_mining(lg, mask)
                    tag = f"{ki}_{k}_{smi}"
                    out[f"loss_{tag}"] = val.detach().numpy().astype(np.float32)
                    out[f"du_{tag}"] = u.grad.numpy()[::gstride].copy()
                    out[f"dv_{tag}"] = v.grad.numpy()[::gstride].copy()
                    if ki == 1:  # the mask does not depend on the loss class
                        out[f"mask_{k}_{smi}"] = np.packbits(mask.numpy())
                        out[f"logits_{smi}"] = lg.numpy()
                        if k == 4:   # hard_mining is defined upstream but never called: pinned through its own output
                            with torch.no_grad():
                                hard = fn.hard_mining(lg, fn.negative_masks(lg, item_idx=item_idx, pos_idx=pos_idx))
                            out[f"hard_{k}_{smi}"] = np.packbits(hard.numpy())
        name = f"losses_B{B}_N{N}_d{d}_P{P}.npz"
        np.savez_compressed(HERE / name, **out)
        print(name, (HERE / name).stat().st_size // 1024, "KiB")


if __name__ == "__main__":
    main()
