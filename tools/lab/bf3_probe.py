"""Time ItemIndex.search(path="bf16") at the bench's retrieval shape (HIP events around 200 calls), with and without
exclusion lists, against the fp32 tile engine; a sweep over Q.

    python tools/lab/bf3_probe.py [Q] [N] [d]
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mf = importlib.import_module("matrix-factorization-torch_amd")

N = int(sys.argv[2]) if len(sys.argv) > 2 else 62423
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
qs = [int(sys.argv[1])] if len(sys.argv) > 1 else [1024, 32, 64, 256, 512, 4096]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
items = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=-1).to(dev)
index = mf.retrieval.ItemIndex(items)
import ctypes
lib = mf._lib.lib()
cnt = (ctypes.c_ulonglong * 3)()
lib.mf_probe_bf3_candidates.argtypes = [ctypes.c_void_p, ctypes.c_int]
for Q in qs:
    q = torch.nn.functional.normalize(torch.randn(Q, d, generator=g), dim=-1).to(dev)
    lens = torch.randint(20, 300, (Q,), generator=g)
    off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(dev)
    ids = torch.randint(0, N, (int(lens.sum()),), generator=g).to(dev)
    for path, csr in (("bf16", None), ("bf16", (off, ids)), ("tiles", (off, ids))):
        if path == "tiles" and Q != qs[0]:
            continue
        for _ in range(20):
            index.search(q, 20, path=path, exclude_csr=csr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            index.search(q, 20, path=path, exclude_csr=csr)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        extra = ""
        if path == "bf16":
            lib.mf_probe_bf3_candidates(None, 1)
            index.search(q, 20, path=path, exclude_csr=csr)
            lib.mf_probe_bf3_candidates(cnt, 0)
            extra = f"  candidates / query {cnt[0] / max(cnt[1], 1):7.1f} (rescored {cnt[2] / max(cnt[1], 1):6.1f})"
        print(f"path {path:6s} excl {'yes' if csr else 'no ':3s} Q {Q:5d} N {N} d {d}: {us:8.1f} us / call  {Q / us:8.2f} M queries/s{extra}", flush=True)
