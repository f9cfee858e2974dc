"""B = 32 steps (the reference's default batch), eager vs captured, InfoNCE and the reference default PairwiseHinge k=4.
    python tools/lab/b32_graph_probe.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
lib = mf._lib.lib()
dev = torch.device("cuda:0")
for loss, k in (("InfomationNoiseContrastiveEstimationLoss", 0), ("PairwiseHingeLoss", 4)):
    for graph in (False, True):
        leg = bench.run_train_leg(mf, lib, dev, batch=32, steps=300, warmup=20, graph=graph, loss=loss, num_negatives=k,
                                  use_logq=(k == 0), spin=False)
        print(f"B=32 {loss[:14]:14s} k={k} {'graph' if graph else 'eager'}: {leg['ms_per_step']:.4f} ms / step", flush=True)
