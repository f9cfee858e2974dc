"""Config C2 (MovieLens-1M shape, d = 64) training leg alone: step time and the sweeps' fractions.
    python tools/lab/c2_probe.py [dim]"""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
lib = mf._lib.lib()
dev = torch.device("cuda:0")
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 64
leg = bench.run_train_leg(mf, lib, dev, batch=8192, steps=60, warmup=8, num_users=6041, num_items=3884, dim=dim, use_logq=False)
roof = bench.train_roofline(leg["spans"], 8192, dim, 1, "adam")
print(f"d={dim}: {leg['ms_per_step']:.4f} ms/step", json.dumps({k: roof[k] for k in ("all_kernels_avg_ms", "all_sweeps_frac")}))
