"""hipGraph capture of a whole training step.

The reference's default batch is 32 pairs (``BATCH_SIZE``, xfmr_rec/params.py:18).  At that size a step is ~20
launches of a few microseconds each and the Python / HIP launch path -- not the GPU -- sets the step time.  Every
entry point of ``libmf_hip.so`` is capturable (no allocation, no synchronisation, scratch from the caller:
include/mf_hip.h), so the whole step -- tower gathers, loss forward, backward, sparse updates -- can be recorded
once and replayed with one launch.

    step = CapturedStep(trainer.step, example_batch, optimizers=[trainer.opt])
    loss = step(batch)            # copies the batch into the static buffers, replays the graph

The replayed step is bit-identical to the eager one (tests/test_gpu_module.py): same kernels, same order; the only
value a capture would freeze -- Adam's global step -- lives in device memory while an optimizer is ``capturable``.
The warm-up calls the capture needs (lazy initialisation must happen outside it) are REAL training steps on the
example batch.
"""
from __future__ import annotations

from typing import Callable, Mapping, Sequence

import torch


class CapturedStep:
    def __init__(self, fn: Callable[[Mapping[str, torch.Tensor]], torch.Tensor], example_batch: Mapping[str, torch.Tensor],
                 *, optimizers: Sequence[torch.optim.Optimizer] = (), warmup: int = 3) -> None:
        if warmup < 1:
            msg = f"at least one warm-up step is needed before a capture: {warmup = }"
            raise ValueError(msg)
        self.fn = fn
        self.optimizers = list(optimizers)
        self.static = {k: v.clone() for k, v in example_batch.items()}
        for opt in self.optimizers:
            if hasattr(opt, "capturable"):
                opt.capturable = True
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn(self.static)
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn(self.static)
        self.replays = 0

    def __call__(self, batch: Mapping[str, torch.Tensor]) -> torch.Tensor:
        for k, dst in self.static.items():
            src = batch[k]
            if src.shape != dst.shape or src.dtype != dst.dtype:
                msg = f"a captured step replays ONE shape: batch[{k!r}] is {tuple(src.shape)} {src.dtype}, captured {tuple(dst.shape)} {dst.dtype}"
                raise ValueError(msg)
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        for opt in self.optimizers:
            if hasattr(opt, "on_replay"):
                opt.on_replay()
        self.replays += 1
        return self.out
