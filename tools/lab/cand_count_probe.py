"""How many rows pass the bf16 prefilter's bound on TRAINED embeddings (bench.py's retrieval leg)?  Exact fp32 scores by
torch; per query the rows with score >= (k-th best) - 4 eps (the scan's thr sits <= 2 eps below the k-th best approximate
score, which is within 2 eps of the exact one), and the largest number of them inside one chunk / lane half of the scan.
    python tools/lab/cand_count_probe.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
lib = mf._lib.lib()
dev = torch.device("cuda:0")
leg = bench.run_train_leg(mf, lib, dev, batch=8192, steps=100, warmup=10)
tr = leg["trainer"]
with torch.no_grad():
    items = tr.item_matrix()
    q = tr.user_vectors(torch.arange(1, 1025, device=dev))
s = q @ items.T
k = 20
kth = s.topk(k, dim=1).values[:, -1]
eps = 1.01 * (2**-7 + 2**-16 + 128 * 2**-22) * q.norm(dim=1) * items.norm(dim=1).max()
for mult in (2, 4):
    cnt = (s >= (kth - mult * eps)[:, None]).sum(dim=1)
    print(f"rows with score >= kth - {mult} eps: mean {cnt.float().mean():.1f}  median {cnt.float().median():.0f}  max {int(cnt.max())}"
          f"  (eps mean {eps.mean():.5f}, kth mean {kth.mean():.4f}, top1 mean {s.max(dim=1).values.mean():.4f})")
rows_per_chunk = 512            # 16 tiles of 32 rows (Q = 1024: upc = 8 units of 2 tiles)
hit = s >= (kth - 4 * eps)[:, None]
n = items.shape[0]
pad = (rows_per_chunk - n % rows_per_chunk) % rows_per_chunk
hit = torch.nn.functional.pad(hit, (0, pad)).reshape(hit.shape[0], -1, rows_per_chunk)
per_chunk = hit.sum(dim=2)
print(f"largest count inside one 512-row chunk: mean over queries {per_chunk.max(dim=1).values.float().mean():.1f}, max {int(per_chunk.max())}"
      f" (a lane half sees half of a chunk's rows: lists hold 15)")
