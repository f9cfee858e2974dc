"""Kernel timeline of one steady step of the reference's DEFAULT loss (PairwiseHinge, 4 mined negatives) at the C3 shape.
    rocprofv3 --kernel-trace -d out -o m -- python3 tools/lab/mined_timeline.py ; python tools/lab/mined_timeline.py out/...db"""
import sys

if len(sys.argv) > 1:
    import sqlite3
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name,start,end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if r[0].startswith("mined_rows_kernel")]       # (present with and without the bf16 prefilter)
    a, b = idx[60], idx[61]
    prev = None
    for r in rows[a:b]:
        gap = (r[1] - prev) / 1e3 if prev else 0.0
        print(f"{r[0][:70]:70s} dur {(r[2] - r[1]) / 1e3:7.1f} gap {gap:6.1f}")
        prev = r[2]
    print(f"step {(rows[b][1] - rows[a][1]) / 1e3:.1f} us")
    sys.exit(0)
import importlib
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
dev = torch.device("cuda:0")
batches, _ = bench.make_batches(8, 8192, seed=1000, device=dev)
tr = bench.Trainer(mf, dev, "adam", 4, loss="PairwiseHingeLoss")
import time
lib = mf._lib.lib()
for mode in (1, 0, 1):
    lib.mf_set_mining_prefilter(mode)
    for i in range(60):
        tr.step(batches[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        tr.step(batches[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"mined step, prefilter {'on ' if mode else 'off'}: {(t2 - t0) / 200 * 1e6:.1f} us wall clock (200 steps; the host was done after {(t1 - t0) / 200 * 1e6:.1f} us per step)", flush=True)
import ctypes
lib.mf_timing_reset()
lib.mf_timing_enable(1)
for i in range(100):
    tr.step(batches[i % 8])
torch.cuda.synchronize()
lib.mf_timing_enable(0)
for name in (b"mining_prefilter", b"mining_select", b"mining_seed", b"mining_bound", b"mining_items", b"mining_users", b"mining_scan", b"mining_rescore", b"update_rows", b"gather_rows"):
    tot = ctypes.c_double(0.0)
    n = lib.mf_timing_get(name, ctypes.byref(tot))
    if n:
        print(f"  span {name.decode():14s} {tot.value / n * 1e3:8.1f} us (HIP events, {n} spans)", flush=True)
if os.environ.get("MF_MINE_DBG"):
    import ctypes
    if hasattr(lib, "mf_probe_mining_prefilter"):
        import struct
        buf = (ctypes.c_ulonglong * 8)()
        for bi in range(8):
            lib.mf_probe_mining_prefilter(None, 1)
            tr.step(batches[bi])
            torch.cuda.synchronize()
            lib.mf_probe_mining_prefilter(buf, 0)
            half = struct.unpack("f", struct.pack("I", buf[7] & 0xFFFFFFFF))[0]
            print(f"batch {bi}: {buf[0] / max(buf[1], 1):.1f} candidates rescored per user over {buf[1]} users, {buf[2]} users walked exactly "
                  f"(no bound {buf[3]}, hard-class bounds {buf[4]}, hit buffers overflowed {buf[5]}; {buf[6]} entries in overflowing lane lists, largest half {half:.4g})", flush=True)
