import importlib, sys, torch
sys.path.insert(0, "/root/repo")
import bench
mf = importlib.import_module("matrix-factorization-torch_amd")
dev = torch.device("cuda:0")
batches, _ = bench.make_batches(8, 32, seed=1000, device=dev)
tr = bench.Trainer(mf, dev, "adam", 4, loss="PairwiseHingeLoss", use_logq=False)
for i in range(60):
    tr.step(batches[i % 8])
torch.cuda.synchronize()
