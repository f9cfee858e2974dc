"""Step time of bench.py's training step right after start, after an idle gap, and after spin-ups: shows
that the device needs ~25 uninterrupted steps (~30 ms) before the SAME step runs at its steady speed, that
half a second of idling brings the ramp back, and which kind of spin-up does (not) avoid it.

    python tools/ramp_probe.py                 # cold, continued, after idle, after GEMM spins
    python tools/ramp_probe.py lossspin 0.5    # 0.5 s of the loss sweeps on scratch data first, then cold
"""
import importlib
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
dev = torch.device("cuda:0")
batches, _ = bench.make_batches(8, 8192, seed=1000, device=dev)
tr = bench.Trainer(mf, dev, "adam", 0)
host_ms = []


def run(n, tag):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n):
        h0 = time.perf_counter()
        ev[i].record()
        tr.step(batches[i % 8])
        host_ms.append(1e3 * (time.perf_counter() - h0))
    ev[n].record()
    torch.cuda.synchronize()
    print(tag, " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.2f}" for i in range(n)))


def gemm_spin(m, n, seconds):
    x, y = torch.randn(m, 128, device=dev), torch.randn(n, 128, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            (x @ y.T).sum()
        torch.cuda.synchronize()


if len(sys.argv) > 2 and sys.argv[1] == "lossspin":      # spin with the sweeps themselves, on scratch data
    g = torch.Generator(device="cpu").manual_seed(3)
    su = torch.nn.functional.normalize(torch.randn(8192, 128, generator=g), dim=-1).to(dev).requires_grad_()
    sv = torch.nn.functional.normalize(torch.randn(16384, 128, generator=g), dim=-1).to(dev).requires_grad_()
    fn = mf.losses.InfomationNoiseContrastiveEstimationLoss()
    b0 = batches[0]
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < float(sys.argv[2]):
        for _ in range(50):                      # back to back: the host runs ahead, the device never idles
            fn(su, sv, b0["target"], item_idx=b0["item"], pos_idx=b0["pos"]).backward()
        torch.cuda.synchronize()
        n += 50
    print("loss spin iterations:", n)
run(40, "cold (ms per step)          :")
print("host enqueue ms per cold step:", " ".join(f"{x:.2f}" for x in host_ms))
run(12, "continued                   :")
time.sleep(0.5)
run(20, "after 0.5 s idle            :")
gemm_spin(4096, 4096 * 32, 0.2)
run(12, "after 0.2 s of a large GEMM :")
time.sleep(0.5)
gemm_spin(8192, 16384, 0.2)
run(12, "after idle + step-shaped GEMM:")
