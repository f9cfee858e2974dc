"""In-kernel phase stamps of the two bf16 scans (lab build only: make ... EXTRA=-DMF_BF3_LAB, MF_HIP_LIB=.../libmf_hip_lab.so):
per workgroup s_memrealtime at entry / loop start / loop end / exit -> where a 20 us kernel of 8 tiles per workgroup spends its time.

    MF_HIP_LIB=matrix-factorization-torch_amd/lib/libmf_hip_lab.so python tools/lab/bf3_stamps.py [Q] [N] [d]
"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mf = importlib.import_module("matrix-factorization-torch_amd")
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 62423
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
items = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=-1).to(dev)
index = mf.retrieval.ItemIndex(items)
lib = mf._lib.lib()
lib.mf_probe_bf3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
q = torch.nn.functional.normalize(torch.randn(Q, d, generator=g), dim=-1).to(dev)
lens = torch.randint(20, 300, (Q,), generator=g)
off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(dev)
ids = torch.randint(0, N, (int(lens.sum()),), generator=g).to(dev)
for _ in range(30):
    index.search(q, 20, path="bf16", exclude_csr=(off, ids))
lib.mf_probe_bf3_stamps(None, 1)
for _ in range(5):
    index.search(q, 20, path="bf16", exclude_csr=(off, ids))
raw = np.zeros(2 * 4096 * 4 + 4096 * 8 + 8, dtype=np.uint64)
lib.mf_probe_bf3_stamps(raw.ctypes.data_as(ctypes.c_void_p), 1)
buf = raw[: 2 * 4096 * 4].reshape(2, 4096, 4)
fin = raw[2 * 4096 * 4: 2 * 4096 * 4 + 4096 * 8].reshape(4096, 8)[:Q].astype(np.int64)
for ps, name in enumerate(("seed", "scan")):
    st = buf[ps]
    st = st[st[:, 0] > 0].astype(np.int64)
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0
    print(f"{name}: {len(st)} workgroups; entry spread {us[:, 0].max():.2f} us; "
          f"prologue {np.median(us[:, 1] - us[:, 0]):.2f}; loop {np.median(us[:, 2] - us[:, 1]):.2f}; "
          f"tail {np.median(us[:, 3] - us[:, 2]):.2f}; first entry -> last exit {us[:, 3].max():.2f} us; "
          f"median exit {np.median(us[:, 3]):.2f}", flush=True)
if len(sys.argv) <= 4:
    seed0, scan0 = buf[0][buf[0][:, 0] > 0][:, 0].min(), buf[1][buf[1][:, 0] > 0][:, 0].min()
    print(f"seed start -> scan start {(int(scan0) - int(seed0)) / 100.0:.2f} us")
fin = fin[fin[:, 0] > 0]
if len(fin):
    t0 = fin[:, 0].min()
    us = (fin - t0) / 100.0
    names = ("slots gathered", "second cut", "rows landed", "chains done", "top-k done")
    print(f"final: {len(fin)} waves; entry spread {us[:, 0].max():.2f} us (median {np.median(us[:, 0]):.2f}); " +
          "; ".join(f"{nm} +{np.median(us[:, i + 1] - us[:, i]):.2f}" for i, nm in enumerate(names)) +
          f"; first entry -> last exit {us[:, 5].max():.2f} us; median wave {np.median(us[:, 5] - us[:, 0]):.2f} us")
    scan0 = buf[1][buf[1][:, 0] > 0][:, 0].astype(np.int64)
    print(f"scan start -> final start {(t0 - scan0.min()) / 100.0:.2f} us")

# clock probe: shader cycles (s_memtime) across the tile loop against the 100 MHz counter
lib.mf_probe_bf3_stamps(None, 2)
for _ in range(3):
    index.search(q, 20, path="bf16", exclude_csr=(off, ids))
lib.mf_probe_bf3_stamps(raw.ctypes.data_as(ctypes.c_void_p), 1)
buf = raw[: 2 * 4096 * 4].reshape(2, 4096, 4)
for ps, name in enumerate(("seed", "scan")):
    st = buf[ps]
    st = st[st[:, 1] > 0].astype(np.int64)
    us = (st[:, 2] - st[:, 1]) / 100.0
    cyc = st[:, 0] - st[:, 3]
    print(f"{name}: loop {np.median(us):.2f} us = {np.median(cyc):.0f} shader cycles -> {np.median(cyc / us) / 1e3:.2f} GHz")
