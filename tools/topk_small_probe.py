"""Latency of the small-batch retrieval scan (mf_topk_small) on one GPU: Q in {1, 4, 8, 32} queries against the
ML-25M-shaped catalog (62,423 x 128) and a 1 M-row one, with per-query exclusion lists on the device.
Device time from HIP events inside the library (mf_timing, both launches of a call) and wall time of a queued
burst of calls.    python tools/topk_small_probe.py"""
import ctypes
import importlib
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
mf = importlib.import_module("matrix-factorization-torch_amd")
lib = mf._lib.lib()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for n, d in ((62423, 128), (1_000_000, 128), (62423, 64)):
    items = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1).to(dev)
    index = mf.retrieval.ItemIndex(items)
    index.blocked()
    for q in (1, 4, 8, 32):
        queries = torch.nn.functional.normalize(torch.randn(q, d, generator=g), dim=-1).to(dev)
        lens = torch.randint(20, 300, (q,), generator=g)
        off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(dev)
        ids = torch.randint(0, n, (int(lens.sum()),), generator=g).to(dev)
        for csr in (None, (off, ids)):
            for _ in range(20):
                index.search(queries, 20, exclude_csr=csr, path="scan")
            torch.cuda.synchronize()
            lib.mf_timing_reset()
            lib.mf_timing_enable(1)
            reps = 200
            t0 = time.perf_counter()
            for _ in range(reps):
                index.search(queries, 20, exclude_csr=csr, path="scan")
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps
            lib.mf_timing_enable(0)
            tot = ctypes.c_double(0.0)
            cnt = lib.mf_timing_get(b"topk_small", ctypes.byref(tot))
            us = tot.value / max(cnt, 1) * 1e3
            print(f"N={n} d={d} Q={q:2d} excl={'yes' if csr else 'no ':3s}: device {us:7.2f} us/call  "
                  f"({n * d * 4 / max(us, 1e-9) / 1e3:7.1f} GB/s of catalog), wall {wall * 1e6:7.1f} us/call", flush=True)
