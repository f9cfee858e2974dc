"""Lab: which rows differ between the one-launch step and the multi-kernel step (config of the failing parity test)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mf = importlib.import_module("matrix-factorization-torch_amd")
DEV = "cuda:0"
opt_name, kind, k, b, d = "adam", "PairwiseHingeLoss", 4, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_users, n_items = 60, 90
g = torch.Generator().manual_seed(b * 7 + d)
def make():
    towers = mf.models.init_towers(mf.models.ModelConfig(num_users=n_users, num_items=n_items, hidden_size=d), device=DEV)
    return towers, mf.optim.RowAdam(towers.parameters(), lr=0.05)
ta, oa = make(); tb, ob = make()
with torch.no_grad():
    for name in ("user", "item"): tb[name].weight.copy_(ta[name].weight)
fn = getattr(mf.losses, kind)(num_negatives=k, sigma=1.3, margin=0.7)
lists = [torch.randperm(n_items - 1, generator=g)[: int(ln)] + 1 for ln in torch.randint(0, 12, (n_users,), generator=g)]
fused = mf.fused.FusedSmallStep(tb, ob, fn)
one = torch.ones((), device=DEV)
for step in range(3):
    user = torch.randint(1, n_users, (b,), generator=g); item = torch.randint(1, n_items, (2 * b,), generator=g)
    user[1] = user[0]; item[b] = item[0]; item[2] = item[3]
    target = torch.randint(-1, 6, (b,), generator=g)
    batch = {"user": user.to(DEV), "item": item.to(DEV), "target": target.to(DEV)}
    pos = torch.randint(0, n_items, (b, 6), generator=g); pos[:, 0] = item[:b]; batch["pos"] = pos.to(DEV)
    u = ta["user"](batch["user"]); v = ta["item"](batch["item"])
    want = fn(u, v, batch["target"], item_idx=batch["item"], pos_idx=batch["pos"])
    want.backward(one); oa.step(); oa.zero_grad(set_to_none=True)
    got = fused(batch)
    print("step", step, float(got), float(want))
    for name, ids in (("user", user), ("item", item)):
        diff = (ta[name].weight - tb[name].weight).abs().amax(1).cpu()
        bad = torch.nonzero(diff > 0).flatten().tolist()
        cnt = {r: int((ids == r).sum()) for r in bad}
        print("  ", name, "rows differing:", [(r, f"{float(diff[r]):.3g}", "x%d" % cnt[r]) for r in bad])
    if any((ta[n].weight != tb[n].weight).any() for n in ("user", "item")): break
