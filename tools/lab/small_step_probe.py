"""The reference-default step (B = 32, PairwiseHinge, 4 mined negatives, row-Adam) three ways: eager multi-kernel, one hipGraph
replay, one launch (mf_step_small).  Host time per call and device time (events).

    python tools/lab/small_step_probe.py [B] [dim]
"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
batches, _ = bench.make_batches(8, B, seed=1000, device=dev)
towers = mf.models.init_towers(mf.models.ModelConfig(num_users=bench.NUM_USERS, num_items=bench.NUM_ITEMS, hidden_size=dim), device=dev)
opt = mf.optim.RowAdam(list(towers.parameters()), lr=1e-4)
fn = mf.losses.PairwiseHingeLoss(num_negatives=4)
step = mf.fused.FusedSmallStep(towers, opt, fn)
for i in range(20):
    step(batches[i % 8])
torch.cuda.synchronize()
for reps in (200, 2000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(reps):
        step(batches[i % 8])
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"one launch, {reps} steps: enqueue {1e6 * (t1 - t0) / reps:7.1f} us / step, total {1e6 * (t2 - t0) / reps:7.1f} us / step, events {1e3 * e0.elapsed_time(e1) / reps:7.1f} us / step")

torch.cuda.synchronize()
st = step._ws[:128].view(torch.int64).cpu().tolist()
names = ["ids + id table", "masks (under the rows' latency)", "rows + normalise", "norms", "logits + mining + rows", "losses", "backward", "both updates"]
print("phases of the last step (us): " + ", ".join(f"{n} {(st[i + 1] - st[i]) / 100:.1f}" for i, n in enumerate(names)) + f"; total {(st[len(names)] - st[0]) / 100:.1f}")
