"""Does a hipGraph make the step's independent branches overlap?  B = 8192 headline step: eager, captured, captured with
the hit masks built on a forked stream (MF_BENCH_PREPARE=1) and with the two table updates on two streams
(MF_TABLE_STREAMS=1).  Eager cross-stream joins cost 20-30 us each on this part; inside a graph the
executor resolves the dependencies.    python tools/graph_probe.py"""
import importlib
import os
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

mf = importlib.import_module("matrix-factorization-torch_amd")
lib = mf._lib.lib()
dev = torch.device("cuda:0")
for name, env, graph in (("eager", {}, False), ("graph", {}, True), ("graph+masks forked", {"MF_BENCH_PREPARE": "1"}, True),
                         ("graph+masks forked+2 update streams", {"MF_BENCH_PREPARE": "1", "MF_TABLE_STREAMS": "1"}, True),
                         ("eager+masks forked+2 update streams", {"MF_BENCH_PREPARE": "1", "MF_TABLE_STREAMS": "1"}, False)):
    for k in ("MF_BENCH_PREPARE", "MF_TABLE_STREAMS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    leg = bench.run_train_leg(mf, lib, dev, batch=8192, steps=100, warmup=10, graph=graph)
    print(f"{name:42s}: {leg['ms_per_step']:.4f} ms / step", flush=True)
    del leg
