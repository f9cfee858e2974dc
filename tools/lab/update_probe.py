"""Time mf_update_adam alone on the bench's id distributions (Zipf item ids + uniform negatives; log-normal users).

    python tools/lab/update_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib

mf = importlib.import_module("matrix-factorization-torch_amd")
import bench  # noqa: E402

dev = torch.device("cuda:0")
lib = mf._lib.lib()
d = 128
batches, _ = bench.make_batches(4, 8192, seed=1000, device=dev)
cases = {"user 8192 (log-normal)": (bench.NUM_USERS, [b["user"] for b in batches]),
         "item 16384 (zipf + uniform)": (bench.NUM_ITEMS, [b["item"] for b in batches]),
         "item 16384 uniform": (bench.NUM_ITEMS, [torch.randint(1, bench.NUM_ITEMS, (16384,), device=dev) for _ in range(4)]),
         "32 ids": (bench.NUM_ITEMS, [torch.randint(1, bench.NUM_ITEMS, (32,), device=dev) for _ in range(4)])}
for name, (rows, ids) in cases.items():
    n = ids[0].numel()
    table = torch.randn(rows, d, device=dev)
    m, v = torch.zeros_like(table), torch.zeros_like(table)
    grad = torch.randn(n, d, device=dev)
    ws = mf._lib.workspace(lib.mf_update_ws_bytes(n, d), dev)
    uniq = sum(int(torch.unique(i).numel()) for i in ids) / len(ids)
    top = max(int(torch.bincount(i).max()) for i in ids)

    def call(i, step):
        mf._lib.check(lib.mf_update_adam(table.data_ptr(), m.data_ptr(), v.data_ptr(), rows, d, ids[i % 4].data_ptr(), n, grad.data_ptr(),
                                         0, step, None, 1e-4, 0.9, 0.999, 1e-8, 0.0, ws.data_ptr(), ws.numel(), None))
    for i in range(20):
        call(i, i + 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200):
        call(i, i + 21)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    moved = (n * d * 4 + uniq * d * 4 * 6) / 1e9
    print(f"{name:30s} n {n:6d} unique {uniq:8.0f} largest run {top:5d}: {us:7.1f} us / call  ({moved / (us * 1e-6):7.0f} GB/s algorithmic)")
