import importlib, sys, torch
sys.path.insert(0, "/root/repo")
mf = importlib.import_module("matrix-factorization-torch_amd")
g = torch.Generator().manual_seed(0)
items = torch.nn.functional.normalize(torch.randn(62423, 128, generator=g), dim=-1).cuda()
index = mf.retrieval.ItemIndex(items)
for q in (1, 8, 32):
    qs = torch.nn.functional.normalize(torch.randn(q, 128, generator=g), dim=-1).cuda()
    for _ in range(100):
        index.search(qs, 20)
torch.cuda.synchronize()
