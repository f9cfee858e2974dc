"""Lab: the one-launch step with CSR positives where one user of the batch holds 30,000 positives."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mf = importlib.import_module("matrix-factorization-torch_amd")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
nu, ni, d, b = 5000, 62423, 32, 32
towers = mf.models.init_towers(mf.models.ModelConfig(num_users=nu, num_items=ni, hidden_size=d), device=dev)
opt = mf.optim.RowAdam(list(towers.parameters()), lr=1e-4)
fn = mf.losses.PairwiseHingeLoss(num_negatives=4)
step = mf.fused.FusedSmallStep(towers, opt, fn)
for heavy in (20, 2000, 30000):
    lens = torch.randint(5, 40, (nu,), generator=g)
    lens[7] = heavy
    off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(dev)
    items = torch.randint(1, ni, (int(lens.sum()),), generator=g).to(dev)
    user = torch.randint(1, nu, (b,), generator=g); user[3] = 7
    batch = {"user": user.to(dev), "item": torch.randint(1, ni, (2 * b,), generator=g).to(dev), "target": torch.ones(b, device=dev)}
    batch["pos_csr"] = (batch["user"], off, items)
    for _ in range(20):
        step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        step(batch)
    torch.cuda.synchronize()
    print(f"heaviest list {heavy:6d}: {(time.perf_counter() - t0) / 300 * 1e6:8.1f} us / step", flush=True)
