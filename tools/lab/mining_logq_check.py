"""Mined losses WITH logQ on an item axis long enough for the seeding pass: oracle vs fp32 search vs bf16 prefilter."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import chain, losses as ol  # noqa: E402
from tests import test_gpu_parity as tp  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
lib = mf._lib.lib()
g = torch.Generator().manual_seed(3)
for (b, n, d, k, sigma) in ((512, 2560, 64, 4, 1.0), (300, 4100, 128, 7, 1.0)):
    t = tp._random_case(b, n, d, 6, seed=b + n, n_items=n // 3)
    logq = torch.log(torch.rand(n, generator=g) * 0.9 + 0.05)
    lg = chain.logits(t["u"].numpy(), t["v"].numpy(), t["target"].numpy(), sigma, logq.numpy())
    for kind in ("PairwiseHingeLoss", "InfomationNoiseContrastiveEstimationLoss"):
        want = float(ol.loss(kind, t["u"], t["v"], t["target"], item_idx=t["item_idx"], pos_idx=t["pos_idx"], num_negatives=k, sigma=sigma,
                             margin=0.5, logq=logq, mining_logits=lg))
        out = []
        for mode in (0, 2):
            lib.mf_set_mining_prefilter(mode)
            out.append(tp._run_gpu(mf, kind, t, k, sigma, 0.5, logq)[0])
        lib.mf_set_mining_prefilter(1)
        print(f"b={b} n={n} d={d} k={k} {kind}: oracle {want:.6f}  fp32 search {out[0]:.6f}  prefilter {out[1]:.6f}", flush=True)
