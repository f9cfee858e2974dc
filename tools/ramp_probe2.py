"""Is the slow start of the training step tied to the device having idled, or to the trainer?  Warm trainer ->
scratch loss sweeps back to back (no idle) -> trainer steps again (same process, every kernel already loaded)."""
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


def run(n, tag, trainer=tr):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for i in range(n):
        ev[i].record()
        trainer.step(batches[i % 8])
    ev[n].record()
    torch.cuda.synchronize()
    print(tag, " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.2f}" for i in range(n)))


g = torch.Generator(device="cpu").manual_seed(3)
su = torch.nn.functional.normalize(torch.randn(8192, 128, generator=g), dim=-1).to(dev).requires_grad_()
sv = torch.nn.functional.normalize(torch.randn(16384, 128, generator=g), dim=-1).to(dev).requires_grad_()
fn = mf.losses.InfomationNoiseContrastiveEstimationLoss()
b0 = batches[0]


def scratch(n):
    for _ in range(n):
        fn(su, sv, b0["target"], item_idx=b0["item"], pos_idx=b0["pos"]).backward()


run(50, "trainer, cold           :")
scratch(150)                      # ~0.17 s of sweeps, queued behind the trainer's last step: no idle
run(12, "same trainer after scratch sweeps, no idle:")
tr2 = bench.Trainer(mf, dev, "adam", 0)     # fresh tables / optimizer state, built while the queue is still full?
scratch(150)
run(20, "FRESH trainer right after scratch sweeps  :", tr2)
torch.cuda.synchronize()
time.sleep(0.3)
run(20, "same (now warm) trainer after 0.3 s idle  :", tr2)
