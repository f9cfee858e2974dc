"""Step time of the reference's DEFAULT training loss (PairwiseHingeLoss, num_negatives = 4: semi-hard mining,
sparse backward) at the bench shape, beside the dense InfoNCE step.  python tools/mined_probe.py"""
import ctypes
import importlib
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
lib = mf._lib.lib()
dev = torch.device("cuda:0")
batches, _ = bench.make_batches(8, 8192, seed=1000, device=dev)
for name, k in (("InfomationNoiseContrastiveEstimationLoss", 0), ("PairwiseHingeLoss", 4), ("PairwiseLogisticLoss", 4),
                ("InfomationNoiseContrastiveEstimationLoss", 32)):
    tr = bench.Trainer(mf, dev, "adam", k)
    tr.loss_fn = getattr(mf.losses, name)(num_negatives=k)
    for i in range(30):
        tr.step(batches[i % 8])
    torch.cuda.synchronize()
    lib.mf_timing_reset()
    lib.mf_timing_enable(1)
    t0 = time.perf_counter()
    for i in range(100):
        tr.step(batches[i % 8])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    lib.mf_timing_enable(0)
    tot = ctypes.c_double(0.0)
    n = lib.mf_timing_get(b"mining_select", ctypes.byref(tot))
    sel = f"  mining_select {tot.value / n:.3f} ms" if n else ""
    if hasattr(lib, "mf_probe_mining_counters") and k:
        buf = (ctypes.c_ulonglong * 16)()
        lib.mf_probe_mining_counters(buf, 1)
        tr.step(batches[0])
        lib.mf_probe_mining_counters(buf, 1)
        w = max(buf[8], 1)
        names = ["cycles", "settle", "-", "compact_slow", "accepted", "body_slices", "compact_calls", "slow_rows"]
        sel += "  per_wave{" + ", ".join(f"{nm}={buf[j] / w:.0f}" for j, nm in enumerate(names) if nm != "-") + f"}} waves={buf[8]}"
    print(f"{name:44s} num_negatives={k:2d}: {1e3 * dt:.3f} ms / step  ({8192 / dt / 1e6:.2f} M pairs/s){sel}")
