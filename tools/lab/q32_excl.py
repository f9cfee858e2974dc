"""Lab: 32 queries with exclusion lists through the scan path (per-kernel times under rocprofv3 --stats)."""
import importlib, pathlib, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
mf = importlib.import_module("matrix-factorization-torch_amd")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n, d, q = 62423, 128, 32
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
items = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=-1).to(dev)
index = mf.retrieval.ItemIndex(items)
qs = torch.nn.functional.normalize(torch.randn(q, d, generator=g), dim=-1).to(dev)
lens = torch.randint(20, 300, (q,), generator=g)
off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(dev)
ids = torch.randint(0, n, (int(lens.sum()),), generator=g).to(dev)
for csr in (None, (off, ids)):
    for _ in range(200):
        index.search(qs, K, exclude_csr=csr, path="scan")
torch.cuda.synchronize()
