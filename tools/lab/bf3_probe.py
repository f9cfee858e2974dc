"""Time ItemIndex.search(path="bf16") at the bench's retrieval shape, per launch (HIP events around 200 calls).
MF_BF3_ABL=1 / 2 / 3 (no staging / no arithmetic / neither: wrong results) prices the parts of the scans.

    python tools/lab/bf3_probe.py [Q] [N] [d]
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mf = importlib.import_module("matrix-factorization-torch_amd")

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 62423
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
items = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=-1).to(dev)
q = torch.nn.functional.normalize(torch.randn(Q, d, generator=g), dim=-1).to(dev)
index = mf.retrieval.ItemIndex(items)
for path in ("bf16", "tiles"):
    for _ in range(20):
        index.search(q, 20, path=path)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        index.search(q, 20, path=path)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    print(f"MF_BF3_ABL={os.environ.get('MF_BF3_ABL', '0')} path {path:6s} Q {Q} N {N} d {d}: {us:8.1f} us / call  {Q / us:8.2f} M queries/s")

lib = mf._lib.lib()
if hasattr(lib, "mf_probe_bf3"):          # a -DBF3_PROBE build (MF_HIP_LIB=...): cycle split of the scans, wave 0 of every workgroup
    import ctypes
    buf = (ctypes.c_ulonglong * 16)()
    lib.mf_probe_bf3.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.mf_probe_bf3(buf, 1)
    index.search(q, 20, path="bf16")
    lib.mf_probe_bf3(buf, 0)
    names = ["total", "prologue", "wait+barrier", "stage issue", "lds+mfma", "epilogue(+rest)", "tail", "workgroups"]
    for ps in range(2):
        v = [buf[ps * 8 + i] for i in range(8)]
        n = max(v[7], 1)
        v[5] = v[0] - v[1] - v[2] - v[3] - v[4]
        print(f"pass {'AB'[ps]}: " + "  ".join(f"{nm} {x / n:9.0f}" for nm, x in zip(names[:6], v[:6])) + f"  (cycles per workgroup, {n} workgroups)")
