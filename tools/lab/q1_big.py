"""Lab: one query against a 1 M-row catalog (per-kernel times under rocprofv3 --stats)."""
import importlib, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
mf = importlib.import_module("matrix-factorization-torch_amd")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n, d = 1_000_000, 128
items = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1).to(dev)
index = mf.retrieval.ItemIndex(items)
q = torch.nn.functional.normalize(torch.randn(1, d, generator=g), dim=-1).to(dev)
for _ in range(200):
    index.search(q, 20, path="scan")
torch.cuda.synchronize()
