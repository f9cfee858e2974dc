"""Host time vs device time of the training step (bench.py's Trainer): how long the Python side needs
to ENQUEUE a step against how long the GPU needs to run it.  If the two are close the step is
host-bound at its seams, whatever the kernels do.

    python tools/step_probe.py [steps]
"""
import importlib
import pathlib
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
batches, _ = bench.make_batches(8, 8192, seed=1000, device=dev)
tr = bench.Trainer(mf, dev, "adam", 0)
cold = len(sys.argv) > 2 and sys.argv[2] == "cold"       # cold: no warm-up at all, per-step times from the first step
if cold:
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for i in range(steps):
        ev[i].record()
        tr.step(batches[i % 8])
    ev[steps].record()
    torch.cuda.synchronize()
    print("cold per-step ms:", " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.3f}" for i in range(steps)))
    sys.exit(0)
for i in range(5):
    tr.step(batches[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    tr.step(batches[i % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / steps:.3f} ms/step   total {1e3 * (t2 - t0) / steps:.3f} ms/step   "
      f"(host ahead of the device by {1e3 * (t2 - t1):.2f} ms at the end of {steps} steps)")

# per-step device time (events at the step seams): shows warm-up transients of the device itself
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
torch.cuda.synchronize()
for i in range(steps):
    ev[i].record()
    tr.step(batches[i % 8])
ev[steps].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
print("per-step ms:", " ".join(f"{x:.3f}" for x in ms))
