"""The mined step at other widths / num_negatives, candidate search through the bf16 prefilter vs the fp32 search (wall clock, eager)."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
lib = mf._lib.lib()
dev = torch.device("cuda:0")
for (dim, k, users, items, B) in [c for c in ((128, 4, bench.NUM_USERS, bench.NUM_ITEMS, 8192), (128, 8, bench.NUM_USERS, bench.NUM_ITEMS, 8192), (128, 16, bench.NUM_USERS, bench.NUM_ITEMS, 8192), (128, 32, bench.NUM_USERS, bench.NUM_ITEMS, 8192),
                                  (64, 4, 6041, 3884, 8192), (128, 4, bench.NUM_USERS, bench.NUM_ITEMS, 4096), (128, 4, bench.NUM_USERS, bench.NUM_ITEMS, 2048)) if not os.environ.get('ONLY_K') or c[1] == int(os.environ['ONLY_K'])]:
    batches, _ = bench.make_batches(8, B, seed=1000, device=dev, num_users=users, num_items=items)
    out, trained = [], []
    for mode in (2, 0):                                       # (2: the prefilter wherever it can serve; the default serves B >= 4096, k <= 8)
        lib.mf_set_mining_prefilter(mode)
        tr = bench.Trainer(mf, dev, "adam", k, loss="PairwiseHingeLoss", num_users=users, num_items=items, dim=dim)
        for i in range(40):
            tr.step(batches[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(100):
            tr.step(batches[i % 8])
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 100 * 1e6)
        trained.append(tr)
    lib.mf_set_mining_prefilter(1)
    extra = ""
    if os.environ.get("MF_MINE_DBG") and hasattr(lib, "mf_probe_mining_prefilter"):
        import ctypes
        buf = (ctypes.c_ulonglong * 8)()
        lib.mf_probe_mining_prefilter(None, 1)
        lib.mf_set_mining_prefilter(2)
        trained[0].step(batches[0])
        torch.cuda.synchronize()
        lib.mf_probe_mining_prefilter(buf, 0)
        lib.mf_set_mining_prefilter(1)
        extra = f"  [{buf[0] / max(buf[1], 1):.1f} columns rescored per user over {buf[1]} users; spilled {buf[6]}, no bound {buf[3]}, hit buffers overflowed {buf[5]}]"
    print(f"d={dim:3d} num_negatives={k:2d} B={B} tables {users}x{items}: prefilter {out[0]:7.1f} us / step, fp32 search {out[1]:7.1f} us / step{extra}", flush=True)
