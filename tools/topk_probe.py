"""Kernel-level probe of the retrieval select kernel (HIP-event time of `topk_select`).

    MF_HIP_LIB=matrix-factorization-torch_amd/lib/variant.so python tools/topk_probe.py [Q] [N] [d] [k]

Used for A/B builds (csrc/Makefile: BUILD= LIB= EXTRA=-D...).  Prints one line per run; with a
library built with -DMF_PROBE it also prints the number of candidate keys the select kernel kept.
"""
import ctypes
import importlib
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
mf = importlib.import_module("matrix-factorization-torch_amd")
lib = mf._lib.lib()

Q, N, d, k = (int(a) for a in (sys.argv[1:5] + ["1024", "62423", "128", "20"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
q = torch.nn.functional.normalize(torch.randn(Q, d, generator=g), dim=-1).to(dev)
items = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=-1).to(dev)
index = mf.retrieval.ItemIndex(items)
for _ in range(3):
    index.search(q, k)
lib.mf_timing_reset()
lib.mf_timing_enable(1)
for _ in range(20):
    s, i = index.search(q, k)
torch.cuda.synchronize()
tot = ctypes.c_double(0.0)
n = lib.mf_timing_get(b"topk_select", ctypes.byref(tot))
ms = tot.value / max(n, 1)
flops = 2.0 * Q * N * d
line = f"lib={mf._lib.LIB_PATH.name} Q={Q} N={N} d={d} k={k} select_ms={ms:.4f} TFLOPs={flops / ms / 1e9:.1f}"
if hasattr(lib, "mf_probe_topk_cand"):
    lib.mf_probe_topk_cand.restype = ctypes.c_longlong
    line += f" candidates_per_query={lib.mf_probe_topk_cand() / Q:.1f}"
if hasattr(lib, "mf_probe_sel_counters"):
    buf = (ctypes.c_ulonglong * 16)()
    lib.mf_probe_sel_counters(buf, 1)
    index.search(q, k)
    lib.mf_probe_sel_counters(buf, 1)
    w = max(buf[8], 1)
    names = ["cycles", "settle", "warm", "filter_slow", "accepted", "body_slices", "filter_calls", "slow_rows"]
    line += " per_wave{" + ", ".join(f"{nm}={buf[j] / w:.0f}" for j, nm in enumerate(names)) + f"}} waves={buf[8]}"
ref = torch.topk(q @ items.T, k, dim=1)
line += f" idx_equal_torch={bool((ref.indices == i).float().mean() > 0.999)}"
print(line)
