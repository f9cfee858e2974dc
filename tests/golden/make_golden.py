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

Two further families (round 4; the six ``losses_*`` files above are generated exactly as before):

* ``wide_*.npz`` -- the ends of the hyper-parameter range the reference's own tuners draw from
  (xfmr_rec/ray.py:147-149, xfmr_rec/flaml.py:73-79: sigma in [1, 1000), margin in [-1, 1], num_negatives in 2^0..2^5):
  (sigma, margin) in {(30, -0.5), (1000, 1.0)} x num_negatives in {1, 4, 32} (+ 0 on the small shape) on the inputs of
  the 32 x 64 and 256 x 512 shapes.
* ``step_B32_N64_d32_P16.npz`` -- a TABLE-CONSISTENT training step at the reference's default shape (params.py:18,
  lightning.py:33,38-39): raw (un-normalised) tables U, V; ``u = normalize(U[user])``, ``v = normalize(V[item])`` so
  duplicate ids share a row; the reference's seven losses (num_negatives = 4, sigma = 1, margin = 1), d loss / d u and
  d loss / d v of each, and -- for each loss -- the tables after ONE step of ``torch.optim.SGD`` (lr 0.1) and of
  ``torch.optim.AdamW`` (lr 0.05, weight_decay 0) applied to the gradient autograd carries back through
  ``F.normalize`` and the row gather to the tables.
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


WIDE_SIGMA_MARGIN = ((30.0, -0.5), (1000.0, 1.0))
WIDE_SHAPES = ((2, (0, 1, 4, 32)), (5, (1, 4, 32)))          # (index into SHAPES, num_negatives)


def make_wide(ref) -> None:
    for si, ks in WIDE_SHAPES:
        B, N, d, P, gstride, edge = SHAPES[si]
        u0, v0, target, item_idx, pos_idx = make_inputs(B, N, d, P, edge, seed=1234 + si)      # the losses_* inputs
        out = {"u": u0.numpy(), "v": v0.numpy(), "target": target.numpy(), "item_idx": item_idx.numpy(),
               "pos_idx": pos_idx.numpy(), "gstride": np.int64(gstride), "ks": np.asarray(ks, dtype=np.int64),
               "sigma_margin": np.asarray(WIDE_SIGMA_MARGIN, dtype=np.float64)}
        for smi, (sigma, margin) in enumerate(WIDE_SIGMA_MARGIN):
            for k in ks:
                for ki, kind in enumerate(KINDS):
                    fn = getattr(ref, kind)(num_negatives=k, sigma=sigma, margin=margin)
                    u = u0.clone().requires_grad_()
                    v = v0.clone().requires_grad_()
                    val = fn(u, v, target, item_idx=item_idx, pos_idx=pos_idx)
                    val.backward()
                    tag = f"{ki}_{k}_{smi}"
                    out[f"loss_{tag}"] = val.detach().numpy().astype(np.float32)
                    out[f"du_{tag}"] = u.grad.numpy()[::gstride].copy()
                    out[f"dv_{tag}"] = v.grad.numpy()[::gstride].copy()
                with torch.no_grad():
                    lg = -ref.squared_distance(u0, v0) * target.sign()[:, None] * sigma
                    mask = fn.semi_hard_mining(lg, fn.negative_masks(lg, item_idx=item_idx, pos_idx=pos_idx))
                out[f"mask_{k}_{smi}"] = np.packbits(mask.numpy())
            out[f"logits_{smi}"] = lg.numpy()
        name = f"wide_B{B}_N{N}_d{d}_P{P}.npz"
        np.savez_compressed(HERE / name, **out)
        print(name, (HERE / name).stat().st_size // 1024, "KiB")


def make_step(ref) -> None:
    B, N, d, P, k = 32, 64, 32, 16, 4
    n_users, n_items = 41, 37                     # rows 0 are the padding rows (ids are 1-based, prepare.py:85)
    g = torch.Generator().manual_seed(4321)
    U = torch.randn(n_users, d, generator=g) * (0.5 + torch.rand(n_users, 1, generator=g))
    V = torch.randn(n_items, d, generator=g) * (0.5 + torch.rand(n_items, 1, generator=g))
    user = torch.randint(1, n_users, (B,), generator=g)
    user[5] = user[4]                              # the same user twice in a batch
    item = torch.randint(1, n_items, (N,), generator=g)      # 64 draws from 36 ids: duplicates everywhere
    item[B] = item[0]                              # a sampled negative that IS row 0's positive
    target = torch.randint(1, 6, (B,), generator=g)
    target[1], target[2] = 0, -3
    pos_idx = torch.zeros(B, P, dtype=torch.long)
    for i in range(B):
        n_pos = int(torch.randint(1, P + 1, (1,), generator=g))
        extra = torch.randint(1, n_items, (n_pos,), generator=g)
        extra[0] = item[i]
        pos_idx[i, :n_pos] = extra
    out = {"U": U.numpy(), "V": V.numpy(), "user": user.numpy(), "item": item.numpy(), "target": target.numpy(),
           "pos_idx": pos_idx.numpy(), "num_negatives": np.int64(k), "lr_sgd": np.float64(0.1), "lr_adam": np.float64(0.05)}
    for ki, kind in enumerate(KINDS):
        fn = getattr(ref, kind)(num_negatives=k, sigma=1.0, margin=1.0)
        for opt_name in ("sgd", "adam"):
            Up, Vp = torch.nn.Parameter(U.clone()), torch.nn.Parameter(V.clone())
            opt = (torch.optim.SGD([Up, Vp], lr=0.1) if opt_name == "sgd"
                   else torch.optim.AdamW([Up, Vp], lr=0.05, weight_decay=0.0))
            u = torch.nn.functional.normalize(Up[user], dim=-1)
            v = torch.nn.functional.normalize(Vp[item], dim=-1)
            u.retain_grad()
            v.retain_grad()
            val = fn(u, v, target, item_idx=item, pos_idx=pos_idx)
            val.backward()
            if opt_name == "sgd":
                out[f"loss_{ki}"] = val.detach().numpy().astype(np.float32)
                out[f"du_{ki}"] = u.grad.numpy().copy()
                out[f"dv_{ki}"] = v.grad.numpy().copy()
                out[f"dU_{ki}"] = Up.grad.numpy().copy()
                out[f"dV_{ki}"] = Vp.grad.numpy().copy()
            opt.step()
            out[f"U_{opt_name}_{ki}"] = Up.detach().numpy().copy()
            out[f"V_{opt_name}_{ki}"] = Vp.detach().numpy().copy()
    name = f"step_B{B}_N{N}_d{d}_P{P}.npz"
    np.savez_compressed(HERE / name, **out)
    print(name, (HERE / name).stat().st_size // 1024, "KiB")


def main() -> None:
    ref = load_reference()
    torch.set_num_threads(1)
    make_wide(ref)
    make_step(ref)
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
