"""Mask build with CSR positives (ML-25M-like log-normal list lengths, pair-weighted users) against the padded P = 64 form.

    python tools/lab/csr_probe.py
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mf = importlib.import_module("matrix-factorization-torch_amd")
import bench  # noqa: E402

dev = torch.device("cuda:0")
lib = mf._lib.lib()
B, N, d = 8192, 16384, 128
inter = bench.make_csr_interactions(seed=0)
print(f"users {inter['num_users']}, positives {inter['pos_items'].numel():,}, longest list {int(inter['lens'].max())}, "
      f"mean {float(inter['lens'].double().mean()):.1f}")
sampler = mf.data.DeviceInteractionSampler(inter["pair_user"], inter["pair_item"], inter["pair_target"], inter["pos_off"], inter["pos_items"],
                                           num_items=bench.NUM_ITEMS, batch_size=B, seed=1, device=dev)
batches = [sampler.batch(s) for s in range(4)]
tot = [int((inter["lens"].to(dev)[b["user"]["idx"]]).sum()) for b in batches]
print("positives looked up per batch:", tot)
padded, _ = bench.make_batches(4, B, seed=1000, device=dev)
ws = mf._lib.workspace(lib.mf_loss_ws_bytes(B, N, d, 0, 0), dev)


def timed(fn, reps=200):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def csr(i):
    b = batches[i % 4]
    item = torch.cat([b["item"]["idx"], b["neg_item"]["idx"]]) if False else items_cat[i % 4]
    uid, off, it = b["user"]["pos_csr"]
    mf._lib.check(lib.mf_loss_masks_csr(B, N, d, 0, item.data_ptr(), uid.data_ptr(), off.data_ptr(), it.data_ptr(), off.numel() - 1,
                                        ws.data_ptr(), ws.numel(), None))


def pad(i):
    b = padded[i % 4]
    mf._lib.check(lib.mf_loss_masks(B, N, d, 64, 0, b["item"].data_ptr(), b["pos"].data_ptr(), ws.data_ptr(), ws.numel(), None))


items_cat = [torch.cat([b["item"]["idx"], b["neg_item"]["idx"]]).contiguous() for b in batches]
print(f"mask build, padded P = 64 : {timed(pad):7.1f} us")
print(f"mask build, CSR lists     : {timed(csr):7.1f} us")
