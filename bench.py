#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: train pairs/s + full-catalog top-k queries/s.

Workload (BASELINE.json configs[2], "C3"): MovieLens-25M *shape* -- 162,541 users x
62,423 items, d = 128, InfoNCE + logQ correction, synthetic ids (Zipf item popularity,
log-normal user activity, seed 0; no MovieLens files exist offline), random-init
unit-norm tables.

One training "step" = one batch of B (user, positive item, sampled negative item)
triples through: tower gathers -> fused score/loss forward -> backward -> sparse
row update of both tables.  One retrieval "step" = Q queries against the whole
catalog with per-query exclusion lists, k = 20.  Inputs are resident in HBM when the
timed region starts.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment: this process starts the N ranks itself, as FRESH child
processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), before it has made any GPU call, waits for them
and exits with their worst code.  Under ``torch.distributed.run`` the ranks exist already; a WORLD_SIZE that differs from
``--gpus`` is an error (exit 2), never a silently smaller job.  The line reports ``n_gpus``, ``rccl_ranks`` (the size of the
C-side RCCL communicator, ``mf_comm_world``) and ``transport`` ("mf_comm": RCCL called from libmf_hip.so on the compute
stream; "torch": torch.distributed's collectives).  ``MF_BENCH_DRY_RUN=1`` rehearses the launch / barrier / max-over-ranks /
JSON plumbing on CPU (gloo, a stub step, no kernels): tests/test_host_cpu.py.

Whenever this part has idled for a few milliseconds the three sweeps run 13 % slower and recover over ~25 launches
(constant sclk / mclk; tools/ramp_probe.py, ramp_probe2.py, clock_probe.py).  `spin_up` therefore queues ~60 ms of
the leg's dominant kernels on scratch data right in front of the W warm-up steps (no benchmark state is touched),
and the optimizer's moment tables are allocated at construction; K = 20 / W = 3 then reads the same 1.16 ms per step
as K = 100 (without it: 1.26).  Defaults: K = 100, W = 10.

Prints ONE JSON line (rank 0).  `value` is train pairs/s over all ranks; the top-k
leg is reported beside it (`topk`), each with the roofline of its dominant kernel,
and the CPU baseline (the oracle's restatement of the same step, torch CPU, all host
cores, bounded sample) on rank 0 at N = 1.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import importlib
import json
import os
import pathlib
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

NUM_USERS, NUM_ITEMS, DIM = 162_541, 62_423, 128       # ML-25M shape
TIME_EVERY = int(os.environ.get("MF_BENCH_TIME_EVERY", "5"))   # HIP-event pairs around every 5th launch of each timed kernel, inside the timed region
                    # (an event pair costs ~6 us of stream time around the kernel it brackets; an ODD period, so that the two
                    # update launches of a step -- user table, item table -- are both sampled; every 16th measured the step 1.5 us
                    # faster but left the spans noisier)


def time_every(launches: int) -> int:
    return max(1, TIME_EVERY)


PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (MI355X_MICROARCH.md)
LDS_DMA_CHIP_GBS = 6400.0          # measured chip-wide LDS-DMA fill rate (MI355X_MICROARCH.md, 'ldsdma-fill')
PEAK_F32_MFMA_TFLOPS = 157.3                            # MI355X_MICROARCH.md, fp32 matrix
PEAK_HBM_GBS = 8000.0
TOP_K, POS_PAD = 20, 64
REPS = int(os.environ.get("MF_BENCH_REPS", "5"))      # repetitions of every timed region (K steps each): median / min / max
CPU_MAX_THREADS = 64                                  # thread counts the CPU baseline sweeps up to (256-thread hosts oversubscribe)
CPU_BUDGET_S = 15.0                                   # time box of the CPU training sweep


def csrc_sha() -> str:
    """Fingerprint of the kernel sources: PMC traffic figures are only quoted for the sources they were measured on."""
    h = hashlib.sha256()
    pkg = ROOT / "matrix-factorization-torch_amd" / "csrc"
    for f in sorted(list(pkg.glob("*.hip")) + list(pkg.glob("*.h")) + list((ROOT / "include").glob("*.h"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def cpu_model() -> str:
    try:
        for line in pathlib.Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8192, help="pairs per GPU per step")
    ap.add_argument("--queries", type=int, default=1024, help="queries per GPU per retrieval step")
    ap.add_argument("--optimizer", choices=("adam", "sgd"), default="adam")
    ap.add_argument("--num-negatives", type=int, default=0, help="0 = all in-batch negatives (sampled softmax)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline legs (no SURVEY 8(d) matrix)")
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def zipf_weights(n: int, s: float = 1.0) -> torch.Tensor:
    return 1.0 / torch.arange(1, n + 1, dtype=torch.float64) ** s


def make_batches(n_batches: int, batch: int, seed: int, device, *, num_users: int = NUM_USERS, num_items: int = NUM_ITEMS,
                 pos_pad: int = POS_PAD):
    """MovieLens-shaped id batches: Zipf(1) item popularity, log-normal user activity."""
    g = torch.Generator().manual_seed(seed)
    item_w = zipf_weights(num_items - 1)
    user_w = torch.exp(torch.randn(num_users - 1, generator=g, dtype=torch.float64))
    out = []
    for _ in range(n_batches):
        user = torch.multinomial(user_w, batch, replacement=True, generator=g) + 1
        item = torch.multinomial(item_w, batch, replacement=True, generator=g) + 1
        neg = torch.randint(1, num_items, (batch,), generator=g)
        target = torch.randint(1, 6, (batch,), generator=g)
        n_pos = torch.randint(min(8, pos_pad), pos_pad + 1, (batch,), generator=g)
        pos = torch.multinomial(item_w, batch * pos_pad, replacement=True, generator=g).reshape(batch, pos_pad) + 1
        pos[:, 0] = item
        pos[torch.arange(pos_pad)[None, :] >= n_pos[:, None]] = 0
        out.append({k: v.to(device) for k, v in
                    dict(user=user, item=torch.cat([item, neg]), target=target, pos=pos).items()})
    counts = torch.bincount(torch.cat([b["item"].cpu() for b in out]), minlength=num_items)
    return out, counts


def make_csr_interactions(seed: int = 0, num_users: int = NUM_USERS, num_items: int = NUM_ITEMS, mean_len: float = 123.0,
                          sigma: float = 1.25, max_len: int = 30_000) -> dict:
    """MovieLens-25M-shaped interactions WITH the real list-length profile: per-user train-list lengths log-normal
    (>= 16, mean ~ 123 = 0.8 x 153, the heaviest users > 10^4 items), items Zipf(1); every (user, item) of a list is one
    training pair, so a batch draws users in proportion to their list length, as real pairs do."""
    g = torch.Generator().manual_seed(seed)
    mu = torch.log(torch.tensor(mean_len)) - 0.5 * sigma * sigma
    lens = torch.exp(mu + sigma * torch.randn(num_users, generator=g, dtype=torch.float64)).clamp(16, max_len).round().to(torch.int64)
    lens[0] = 0                                               # row 0 is the padding row
    lens[1] = max_len                                         # at least one user at the cap
    total = int(lens.sum())
    items = torch.multinomial(zipf_weights(num_items - 1), total, replacement=True, generator=g) + 1
    off = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)])
    users = torch.repeat_interleave(torch.arange(num_users), lens)
    return {"num_users": num_users, "lens": lens, "pos_off": off, "pos_items": items, "pair_user": users, "pair_item": items,
            "pair_target": torch.randint(1, 6, (total,), generator=g).float()}


def hashed_batches(n_batches: int, batch: int, device, *, users: int, items: int, pos_pad: int = POS_PAD, seed: int = 5):
    """Batches over an id space far larger than any table (config C5: hash / bloom towers): users uniform, items
    log-uniform (~ Zipf(1) without a 100 M-entry weight vector), positives of the same law."""
    import math

    g = torch.Generator().manual_seed(seed)
    zipf = lambda n: (torch.rand(n, generator=g, dtype=torch.float64) * math.log(items)).exp().long().clamp(1, items - 1)  # noqa: E731
    out = []
    for _ in range(n_batches):
        item = zipf(batch)
        pos = zipf(batch * pos_pad).reshape(batch, pos_pad)
        pos[:, 0] = item
        n_pos = torch.randint(min(8, pos_pad), pos_pad + 1, (batch,), generator=g)
        pos[torch.arange(pos_pad)[None, :] >= n_pos[:, None]] = 0
        out.append({k: v.to(device) for k, v in dict(user=torch.randint(1, users, (batch,), generator=g),
                                                     item=torch.cat([item, torch.randint(1, items, (batch,), generator=g)]),
                                                     target=torch.randint(1, 6, (batch,), generator=g), pos=pos).items()})
    return out


def logq_table(device, num_items: int = NUM_ITEMS) -> torch.Tensor:
    """log of the sampling probability of each item row: positives ~ Zipf, negatives ~ uniform."""
    w = zipf_weights(num_items - 1)
    p = 0.5 * w / w.sum() + 0.5 / (num_items - 1)
    return torch.cat([torch.zeros(1, dtype=torch.float64), p.log()]).to(torch.float32).to(device)


class Trainer:
    """The reference's training_step shape (xfmr_rec/lightning.py:97-147,189-192) on the HIP path."""

    def __init__(self, mf, device, optimizer: str, num_negatives: int, *, num_users: int = NUM_USERS,
                 num_items: int = NUM_ITEMS, dim: int = DIM, loss: str = "InfomationNoiseContrastiveEstimationLoss",
                 use_logq: bool = True, num_hashes: int = 0):
        cfg = mf.models.ModelConfig(num_users=num_users, num_items=num_items, hidden_size=dim, num_hashes=num_hashes)
        torch.manual_seed(0)
        self.towers = mf.models.init_towers(cfg, device=device)
        self.loss_fn = getattr(mf.losses, loss)(num_negatives=num_negatives, sigma=1.0)
        params = list(self.towers.parameters())
        self.opt = mf.optim.RowAdam(params, lr=1e-4) if optimizer == "adam" else mf.optim.SparseSGD(params, lr=1e-2)
        if optimizer == "adam":
            self.opt.init_state()              # moment tables allocated here, not inside the first step
        self.dim, self.num_items, self.device = dim, num_items, device
        self.logq = logq_table(device, num_items) if use_logq else None
        self.one = torch.ones((), device=device)      # upstream gradient of loss.backward(): no fill kernel per step

    def item_matrix(self) -> torch.Tensor:
        return self.towers["item"](torch.arange(self.num_items, device=self.device)).detach()

    def user_vectors(self, rows: torch.Tensor) -> torch.Tensor:
        return self.towers["user"](rows).detach()

    def step(self, b) -> torch.Tensor:
        # the hit masks depend on the ids only and can be built on a side stream while the towers gather; measured
        # equal (1.1718 vs 1.1732 ms / step: the cross-stream join costs what the overlap saves), so off by default.
        # (Building the NEXT batch's masks at the start of a step -- a whole step of slack -- also measured equal,
        # 1.0085 vs 1.0088 ms: kernels of a second stream do not slip in beside the sweeps, they queue.)
        masks = (self.loss_fn.prepare_masks(b["item"], b.get("pos"), batch_size=b["user"].numel(), embedding_dim=self.dim,
                                            pos_csr=b.get("pos_csr"))
                 if os.environ.get("MF_BENCH_PREPARE", "0") == "1" else None)
        u = self.towers["user"](b["user"])
        v = self.towers["item"](b["item"])
        # the int64 targets and the logQ table go to the kernel as they are (converted / looked up in its set-up launch);
        # positives: the padded [B, P] matrix, or the producer's CSR lists read in place (b["pos_csr"])
        loss = self.loss_fn(u, v, b["target"], item_idx=b["item"], pos_idx=b.get("pos"), logq_table=self.logq,
                            prepared=masks, pos_csr=b.get("pos_csr"))
        loss.backward(self.one)
        self.opt.step()
        return loss


def spin_up(mf, device, what: str, index=None, *, dim: int = DIM, loss: str = "InfomationNoiseContrastiveEstimationLoss",
            num_negatives: int = 0, batch: int = 8192) -> None:
    """Device spin-up on SCRATCH data, queued right in front of a leg's W warm-up steps (no table, optimizer state or
    batch of the benchmark is touched).  Measured (tools/ramp_probe.py, ramp_probe2.py, clock_probe.py): whenever the
    device has idled for a few milliseconds -- the host preparing batches, an allocation, lazy code loading -- the
    three sweeps run 13 % slower (371 -> 325 us, constant sclk / mclk) and recover over ~25 launches of THIS kind of
    work (a GEMM spin does not do it); a fresh trainer started right behind queued sweeps is at full speed from its
    first step.  So every kernel of the leg is first launched once on tiny inputs (code objects loaded: no stall
    later), then ~60 ms of the leg's dominant kernels are QUEUED, unsynchronised, and the W warm-up steps are issued
    behind them at once: the device never idles between here and the timed region's own synchronize."""
    g = torch.Generator(device="cpu").manual_seed(12345)
    if what == "train":
        tiny = mf.models.init_towers(mf.models.ModelConfig(num_users=512, num_items=512, hidden_size=dim), device=device)
        opt = mf.optim.RowAdam(tiny.parameters(), lr=1e-4)
        fn = getattr(mf.losses, loss)(num_negatives=num_negatives)
        ids = torch.arange(1, 257, device=device)
        item = torch.cat([ids, ids + 200])
        fn(tiny["user"](ids), tiny["item"](item), torch.ones(256, device=device), item_idx=item,
           pos_idx=ids[:, None].repeat(1, POS_PAD), logq=torch.zeros(512, device=device)).backward()
        opt.step()
        u = torch.nn.functional.normalize(torch.randn(batch, dim, generator=g), dim=-1).to(device).requires_grad_()
        v = torch.nn.functional.normalize(torch.randn(2 * batch, dim, generator=g), dim=-1).to(device).requires_grad_()
        item = torch.randint(1, NUM_ITEMS, (2 * batch,), generator=g).to(device)
        pos = torch.randint(0, NUM_ITEMS, (batch, POS_PAD), generator=g).to(device)
        tgt, lq = torch.ones(batch, device=device), torch.zeros(2 * batch, device=device)
        work, n = (lambda: fn(u, v, tgt, item_idx=item, pos_idx=pos, logq=lq).backward()), max(4, int(60 * 8192 / batch) if batch >= 1024 else 8)   # noqa: E731
    else:
        q = torch.nn.functional.normalize(torch.randn(1024, dim, generator=g), dim=-1).to(device)
        work, n = (lambda: index.search(q, TOP_K)), 200   # noqa: E731
    work()
    torch.cuda.synchronize()
    for _ in range(n):                           # queued back to back, NOT synchronised: the warm-up steps follow at once
        work()


def timed(fn, n_steps: int, dist_on: bool) -> float:
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_steps):
        fn(i)
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    return dt


def measured_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/rNN_traffic.json, the latest round: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py).
    The file records the fingerprint of the kernel sources it was measured on: a figure of other sources is not
    quoted (None)."""
    try:
        data = json.loads(sorted((ROOT / "profiles").glob("r*_traffic.json"))[-1].read_text())     # the latest round's
        if data.get("csrc_sha") != csrc_sha():
            return None
        return data["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def kernel_span(lib, name: str):
    tot = ctypes.c_double(0.0)
    n = lib.mf_timing_get(name.encode(), ctypes.byref(tot))
    return (tot.value / n if n else None), n


TRAIN_KERNELS = ("loss_fwd_dense", "loss_bwd_du", "loss_bwd_dv", "mining_select", "mining_prefilter", "gather_rows", "update_rows")


def flat_batch(b) -> dict:
    """A producer's nested batch (InteractionBatchType) as the flat dict the bench trainers take."""
    out = {"user": b["user"]["idx"], "item": torch.cat([b["item"]["idx"], b["neg_item"]["idx"]]), "target": b["target"]}
    if b["user"].get("pos_csr") is not None:
        out["pos_csr"] = (out["user"],) + tuple(b["user"]["pos_csr"][1:])
    else:
        out["pos"] = b["user"]["pos_idx"]
    return out


def spread(ms: list) -> dict:
    srt = sorted(ms)
    return {"n": len(ms), "median": round(srt[len(srt) // 2], 4), "min": round(srt[0], 4), "max": round(srt[-1], 4)}


def run_train_leg(mf, lib, device, *, batch: int, steps: int, warmup: int, optimizer: str = "adam", num_negatives: int = 0,
                  loss: str = "InfomationNoiseContrastiveEstimationLoss", num_users: int = NUM_USERS, num_items: int = NUM_ITEMS,
                  dim: int = DIM, pos_pad: int = POS_PAD, use_logq: bool = True, spin: bool = True, graph: bool = False,
                  seed: int = 1000, reps: int = 1, batches=None, num_hashes: int = 0, id_space=None) -> dict:
    """One single-GPU training leg: fresh tables, `warmup` untimed steps, then `reps` timed regions of `steps` steps each
    (the reported ms / step is the MEDIAN region); HIP-event spans of the dominant kernels every time_every(launches)-th launch.
    graph=True: the step is captured in a hipGraph and replayed.  `batches`: pre-built flat batches (else synthetic padded
    ones); `id_space` = (users, items): ranges the synthetic ids are drawn from when they differ from the table heights
    (hash towers: ids far beyond the bucket counts)."""
    n_batches = min(steps + warmup, 8)
    if batches is None:
        nu, ni = id_space if id_space is not None else (num_users, num_items)
        batches, _ = make_batches(n_batches, batch, seed=seed, device=device, num_users=nu, num_items=ni, pos_pad=pos_pad)
    n_batches = len(batches)
    trainer = Trainer(mf, device, optimizer, num_negatives, num_users=num_users, num_items=num_items, dim=dim, loss=loss,
                      use_logq=use_logq, num_hashes=num_hashes)
    step = trainer.step
    if graph:
        step = mf.graph.CapturedStep(trainer.step, batches[0], optimizers=[trainer.opt], warmup=3)
    if spin:
        spin_up(mf, device, "train", dim=dim, loss=loss, num_negatives=num_negatives, batch=batch)
    for i in range(warmup):
        step(batches[i % n_batches])
    lib.mf_timing_reset()
    if not graph:                                   # (event records inside a replayed graph would time nothing)
        lib.mf_timing_enable(time_every(steps * max(reps, 1)))
    ms = []
    for r in range(reps):
        dt = timed(lambda i, r=r: step(batches[(warmup + r * steps + i) % n_batches]), steps, False)
        ms.append(dt / steps * 1e3)
    lib.mf_timing_enable(0)
    spans = {n: kernel_span(lib, n)[0] for n in TRAIN_KERNELS}
    med = sorted(ms)[len(ms) // 2]
    return {"ms_per_step": med, "pairs_per_s": batch / (med * 1e-3), "reps_ms": ms, "spans": {k: v for k, v in spans.items() if v},
            "trainer": trainer, "batches": batches}


def train_roofline(spans: dict, batch: int, dim: int, world: int, optimizer: str, update_launches: int = 1):
    """roofline of the dominant MFMA sweep + achieved HBM rates of the gather / update kernels (SURVEY 8d figures)."""
    n = 2 * batch
    flops = 2.0 * batch * n * dim                      # one B x N x d contraction per launch
    sweeps = {k: spans[k] for k in ("loss_fwd_dense", "loss_bwd_du", "loss_bwd_dv", "mining_select", "mining_prefilter") if spans.get(k)}
    if not sweeps:
        return None
    dom = max(sweeps, key=sweeps.get)
    ach = flops / (sweeps[dom] * 1e-3) / 1e12
    if dom == "mining_prefilter":
        # The mined losses' candidate search through the split-bf16 prefilter (csrc/mf_mine_bf.h): ONE event pair spans its
        # launches -- item plane + user fragments, fp32 seeding pass over 1/8 of the columns, bound + intervals, the scan on the bf16 cores
        # (three bf16 products per fp32 product + one augmented k-step: 3 + 16/d of the algorithmic flops are EXECUTED), exact
        # rescoring.  `achieved` counts the ALGORITHMIC 2 B N d only and is priced against the bf16 peak, the pipe the dominant
        # kernel runs on; `of_fp32_mfma_peak` is the same figure against the peak an all-fp32 search is bounded by.
        execd = (3.0 + 16.0 / dim) * ach
        out = {"kernel": "mine_items_kernel + select_seed_kernel + mine_bound_kernel + mine_scan_kernel + mine_rescore_kernel",
               "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
               "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": None, "avg_ms": round(sweeps[dom], 4),
               "mfma_dtype": "bf16 x3 split (fp32 accumulate), exact fp32 rescoring", "executed_mfma_frac": round(execd / PEAK_BF16_MFMA_TFLOPS, 4),
               "of_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
               "all_kernels_avg_ms": {k: round(v, 4) for k, v in spans.items()},
               "all_sweeps_frac": {k: round(flops / (v * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) for k, v in sweeps.items()}}
    else:
        out = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
               "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
               "traffic": measured_traffic(dom) if (batch, dim, world) == (8192, 128, 1) else None,
               "avg_ms": round(sweeps[dom], 4),
               "all_kernels_avg_ms": {k: round(v, 4) for k, v in spans.items()},
               "all_sweeps_frac": {k: round(flops / (v * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) for k, v in sweeps.items()}}
    hbm = {}
    if spans.get("gather_rows"):      # two launches per step (B user rows, 2B item rows): 8 n d bytes each (read + write)
        gb = 8.0 * (batch + n) * dim / (2 * spans["gather_rows"] * 1e-3) / 1e9
        hbm["gather_rows"] = {"achieved_GBps": round(gb, 1), "frac_of_8TBps": round(gb / PEAK_HBM_GBS, 4),
                              "algorithmic_bytes_per_step": int(8 * (batch + n) * dim)}
    if spans.get("update_rows"):      # ONE launch per step for both tables since round 4 (mf_update_pair; the sharded trainer: two);
        per_pair = 72 if optimizer == "adam" else 24       # 24 d (SGD) / 72 d (row Adam) bytes per pair
        gb = float(per_pair) * dim * batch / (update_launches * spans["update_rows"] * 1e-3) / 1e9
        hbm["update_rows"] = {"achieved_GBps": round(gb, 1), "frac_of_8TBps": round(gb / PEAK_HBM_GBS, 4),
                              "algorithmic_bytes_per_step": int(per_pair * dim * batch)}
    out["hbm_kernels"] = hbm
    return out


def cpu_train_baseline(batches, logq, optimizer: str, n_steps: int, num_negatives: int):
    """The oracle's restatement of the same step (torch CPU, the current thread setting)."""
    from oracle import embed as oembed, losses as ol

    torch.manual_seed(0)
    ut = torch.randn(NUM_USERS, DIM) / DIM**0.5
    it = torch.randn(NUM_ITEMS, DIM) / DIM**0.5
    state = [torch.zeros_like(ut), torch.zeros_like(ut), torch.zeros_like(it), torch.zeros_like(it)]
    t0 = time.perf_counter()
    for s in range(n_steps):
        b = batches[s % len(batches)]
        ur = ut[b["user"]].requires_grad_()
        ir = it[b["item"]].requires_grad_()
        un = ur / ur.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        vn = ir / ir.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        loss = ol.loss("InfomationNoiseContrastiveEstimationLoss", un, vn, b["target"], item_idx=b["item"],
                       pos_idx=b["pos"], num_negatives=num_negatives, logq=logq[b["item"]])
        loss.backward()
        if optimizer == "adam":
            oembed.adam_update(ut, state[0], state[1], b["user"], ur.grad, step=s + 1, lr=1e-4, weight_decay=0.01)
            oembed.adam_update(it, state[2], state[3], b["item"], ir.grad, step=s + 1, lr=1e-4, weight_decay=0.01)
        else:
            oembed.sgd_update(ut, b["user"], ur.grad, 1e-2)
            oembed.sgd_update(it, b["item"], ir.grad, 1e-2)
    return time.perf_counter() - t0


def cpu_topk_once(q: torch.Tensor, items: torch.Tensor, k: int, excl_rows: torch.Tensor, excl_cols: torch.Tensor):
    """All-core brute force: one matmul, one masked fill (the exclusion lists as two index vectors), one topk."""
    s = q @ items.T
    s[excl_rows, excl_cols] = float("-inf")
    return torch.topk(s, k, dim=1)


def cpu_baselines(args, batches, queries, items, pieces) -> dict:
    """CPU baseline on the host cores of this box (rank 0, N = 1): the oracle's restatement of the SAME workload.  Thread
    counts up to CPU_MAX_THREADS are tried on the FULL batch, one step each, inside a CPU_BUDGET_S time box (a 256-thread
    host is slower oversubscribed than at 32 .. 64 threads, and one step is 2 .. 4 s); the best step is the figure."""
    b = batches[0]["user"].numel()
    cores = os.cpu_count() or 1
    cands = [t for t in (32, 64, 16, 8) if t <= min(cores, CPU_MAX_THREADS)] or [cores]
    lq = logq_table("cpu")
    full = batches[0]
    wb = min(512, b)                                  # a reduced, untimed warm step per thread setting: thread-pool start-up,
    warm = {"user": full["user"][:wb], "item": torch.cat([full["item"][:wb], full["item"][b:b + wb]]),      # first-touch page
            "target": full["target"][:wb], "pos": full["pos"][:wb]}                                            # faults
    sweep, t_start, cut = {}, time.perf_counter(), False
    for t in cands:
        if sweep and time.perf_counter() - t_start > CPU_BUDGET_S:
            cut = True
            break
        torch.set_num_threads(t)
        cpu_train_baseline([warm], lq, args.optimizer, 1, args.num_negatives)
        sweep[t] = cpu_train_baseline(batches[:1], lq, args.optimizer, 1, args.num_negatives)
    best_t = min(sweep, key=sweep.get)
    torch.set_num_threads(best_t)                     # a second timed step at the chosen setting; the better of the two is kept
    at_best = [sweep[best_t], cpu_train_baseline(batches[:1], lq, args.optimizer, 1, args.num_negatives)]
    sweep[best_t] = min(at_best)
    # top-k
    qn = queries.shape[0]
    rows = torch.cat([torch.full((p.numel(),), r, dtype=torch.int64) for r, p in enumerate(pieces)])
    cols = torch.cat(pieces)
    tk = {}
    for t in cands:
        torch.set_num_threads(t)
        cpu_topk_once(queries, items, TOP_K, rows, cols)
        t0 = time.perf_counter()
        for _ in range(3):
            cpu_topk_once(queries, items, TOP_K, rows, cols)
        tk[t] = (time.perf_counter() - t0) / 3
    best_k = min(tk, key=tk.get)
    return {"value": round(b / sweep[best_t], 1), "unit": "pairs/s", "cores": best_t, "kind": "port",
            "cpu_model": cpu_model(), "host_threads_available": cores,
            "thread_sweep_s_per_step": {str(t): round(v, 3) for t, v in sweep.items()},
            "steps_s_at_best": [round(v, 3) for v in at_best], "sweep_cut_by_time_box": cut,
            "sample": f"one step of the same workload (B={b}, N={2 * b}, d={DIM}, InfoNCE+logQ, {args.optimizer}) by oracle/ on "
                      f"torch CPU per thread count in {list(sweep)} (<= {CPU_MAX_THREADS} threads, {CPU_BUDGET_S:.0f} s time box; an untimed "
                      f"{wb}-pair warm step first), two timed steps at the best setting, best kept; top-k: 3 batches of Q={qn} per thread count, best kept",
            "topk_value": round(qn / tk[best_k], 1), "topk_unit": "queries/s", "topk_cores": best_k,
            "topk_thread_sweep_ms": {str(t): round(v * 1e3, 1) for t, v in tk.items()}}


def topk_small_leg(mf, lib, index, device, dim: int, rank: int) -> dict:
    """Q = 1: the reference's own retrieval shape (one query per call, data/lightning.py:237-259).  Device time of a
    call (both launches) from HIP events; bandwidth = catalog bytes / time against HBM."""
    g = torch.Generator().manual_seed(99 + rank)
    out = {}
    for q in (1, 8, 32):
        queries = torch.nn.functional.normalize(torch.randn(q, dim, generator=g), dim=-1).to(device)
        lens = torch.randint(20, 300, (q,), generator=g)
        off = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(device)
        ids = torch.randint(1, index.num_items, (int(lens.sum()),), generator=g).to(device)
        for _ in range(50):
            index.search(queries, TOP_K, exclude_csr=(off, ids), path="scan")
        torch.cuda.synchronize()
        lib.mf_timing_reset()
        lib.mf_timing_enable(1)
        reps = 300
        t0 = time.perf_counter()
        for _ in range(reps):
            index.search(queries, TOP_K, exclude_csr=(off, ids), path="scan")
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        lib.mf_timing_enable(0)
        us = kernel_span(lib, "topk_small")[0] * 1e3
        gbs = index.num_items * dim * 4 / (us * 1e-6) / 1e9
        out[f"q{q}"] = {"latency_us": round(us, 2), "wall_us_per_call": round(wall * 1e6, 1), "queries_per_s": round(q / wall, 1),
                        "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK_HBM_GBS, 4), "frac_of_6.3TBps_achievable": round(gbs / 6300.0, 4),
                        "kernels": ("topk_small_scan_kernel" if q == 1 else "topk_small_mfma_scan_kernel") + " + topk_small_select_kernel", "bound": "hbm"}
    # the same 32 queries through the bf16 prefilter (five launches), which "auto" takes from 33 queries on
    for _ in range(50):
        index.search(queries, TOP_K, exclude_csr=(off, ids), path="bf16")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        index.search(queries, TOP_K, exclude_csr=(off, ids), path="bf16")
    torch.cuda.synchronize()
    out["q32_bf16_prefilter"] = {"wall_us_per_call": round((time.perf_counter() - t0) / 300 * 1e6, 1)}
    return out


def small_step_leg(mf, device, bsz: int, kw: dict, dim: int = DIM) -> dict:
    batches, _ = make_batches(8, bsz, seed=1000, device=device)
    cfg = mf.models.ModelConfig(num_users=NUM_USERS, num_items=NUM_ITEMS, hidden_size=dim)
    torch.manual_seed(0)
    towers = mf.models.init_towers(cfg, device=device)
    opt = mf.optim.RowAdam(list(towers.parameters()), lr=1e-4)
    fn = getattr(mf.losses, kw["loss"])(num_negatives=kw["num_negatives"], sigma=1.0)
    step = mf.fused.FusedSmallStep(towers, opt, fn)
    for i in range(50):
        step(batches[i % 8])
    torch.cuda.synchronize()
    # ten regions of 200 steps, the MEDIAN region reported (a one-off host hiccup right after the hipGraph legs -- tens of
    # milliseconds once -- used to leak into a single 2000-step average: 41 us read 69)
    walls, devs = [], []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(200):
            step(batches[i % 8])
        e1.record()
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / 200)
        devs.append(e0.elapsed_time(e1) / 200)
    wall, dev_ms = sorted(walls)[5], sorted(devs)[5]
    assert step.fallback_steps == 0
    st = step._ws[:128].view(torch.int64).cpu().tolist()         # s_memrealtime (100 MHz) at the phase boundaries of the last step
    names = ("ids_and_id_table", "masks", "rows_normalised", "norms", "logits_mining_rows", "losses", "backward", "both_updates")
    return {"one_launch_ms_per_step": round(wall * 1e3, 4), "one_launch_device_ms_per_step": round(dev_ms, 4),
            "one_launch_device_reps_ms": spread(devs),
            "one_launch_pairs_per_s": round(bsz / wall, 1), "one_launch_kernel": "step_small_kernel (mf_step_small)",
            "one_launch_phases_us": {n: round((st[i + 1] - st[i]) / 100, 1) for i, n in enumerate(names)}}


def spawn_ranks(args) -> int:
    """``--gpus N`` without a launcher: start the N ranks as fresh children of this (GPU-untouched) process."""
    import socket
    import subprocess

    n = args.gpus
    dry = os.environ.get("MF_BENCH_DRY_RUN") == "1"
    if not dry and torch.cuda.device_count() < n:          # (counting devices does not initialise the GPU)
        print(f"bench.py: --gpus {n} but only {torch.cuda.device_count()} visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MF_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, str(pathlib.Path(__file__).resolve()), *sys.argv[1:]], env=env))
    worst, left = 0, list(procs)
    while left:
        for pr in list(left):
            try:
                rc = pr.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            left.remove(pr)
            if rc != 0:
                worst = worst or rc
                for other in left:                         # the others would wait in a collective for ever
                    other.terminate()
    return worst


def restart_on_torch_collectives(why: str) -> None:
    """The C-side RCCL communicator failed its self-test on some rank: RCCL is then in an unknown state in THIS process
    (a mismatched or half-completed operation), so nothing more is attempted here -- no in-process continuation.
    Every rank (``default_comm`` raises on all of them together) agrees on a fresh rendezvous port through the still-working
    torch.distributed group, tears it down, and starts ONE fresh child of itself with ``MF_COMM=torch`` -- torch.distributed's
    own RCCL collectives, a few cross-stream joins slower, never a CPU path -- waits for it and exits with its code.  The child
    of rank 0 prints the JSON line, with ``transport`` = "torch" and a ``comm_note`` saying what happened.  Works the same
    under ``bench.py --gpus N`` (our own parent) and under ``python -m torch.distributed.run`` (the driver's launcher)."""
    import subprocess
    import torch.distributed as dist

    port = int(os.environ.get("MASTER_PORT", "29517")) + 1
    try:
        box = [None]
        if dist.get_rank() == 0:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                box[0] = sk.getsockname()[1]
        dist.broadcast_object_list(box, src=0)
        port = int(box[0])
    except Exception:  # noqa: BLE001  (the agreed fall-back: the old port + 1)
        pass
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
    print(f"bench.py: {why}; restarting this rank in a fresh process on torch.distributed collectives", file=sys.stderr, flush=True)
    env = dict(os.environ, MF_COMM="torch", MASTER_PORT=str(port), MF_BENCH_COMM_NOTE=why[:300])
    sys.exit(subprocess.call([sys.executable, str(pathlib.Path(__file__).resolve()), *sys.argv[1:]], env=env))


def dry_run(args, world: int, rank: int) -> None:
    """The launch contract on CPU (no GPU, no kernels): gloo group, W stub warm-up steps, K stub steps between barriers,
    max over ranks, one JSON line from rank 0 with the fields the driver and the tests read."""
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    if dist.get_world_size() != args.gpus:
        raise SystemExit(2)
    if os.environ.get("MF_BENCH_FAKE_COMM_FAILURE") == "1" and os.environ.get("MF_COMM") != "torch":
        restart_on_torch_collectives("mf_comm self-test failed (simulated: MF_BENCH_FAKE_COMM_FAILURE)")      # the restart path, on CPU
    step = lambda i: time.sleep(0.001)          # noqa: E731
    for i in range(args.warmup):
        step(i)
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ranks = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(ranks)
    if rank == 0:
        print(json.dumps({"metric": "train pairs/sec + full-catalog top-k queries/sec, ML-25M d=128", "value": 0.0, "unit": "pairs/s",
                          "n_gpus": int(ranks), "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t) / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none (dry run)",
                          "config": {"workload": "dry run: launch / barrier / reduction plumbing only", "parallelism": f"dp{world}"},
                          "rccl_ranks": None, "transport": "dry-run (gloo, no kernels)",
                          **({"comm_note": os.environ["MF_BENCH_COMM_NOTE"]} if os.environ.get("MF_BENCH_COMM_NOTE") else {})}), flush=True)
    dist.destroy_process_group()


def main() -> None:
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))             # nothing above has touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("MF_BENCH_DRY_RUN") == "1":
        dry_run(args, world, rank)
        return
    dist_on = world > 1 or os.environ.get("MF_BENCH_FORCE_DIST") == "1"   # the override rehearses the sharded path on one GPU
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.distributed.init_process_group("nccl", device_id=device)   # nccl == RCCL on ROCm
        if torch.distributed.get_world_size() != args.gpus:
            print(f"bench.py: process group of {torch.distributed.get_world_size()} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)
    mf = importlib.import_module("matrix-factorization-torch_amd")
    lib = mf._lib.lib()
    B, Q, K, W = args.batch, args.queries, args.steps, args.warmup
    extras = {}

    # ------------------------------------------------------------------ training leg --
    cold_ms = None
    data_tag, ml_meta = "synthetic", None
    if dist_on:
        n_batches = min(K + W, 8)
        batches, _ = make_batches(n_batches, B, seed=1000 + rank, device=device)
        # default: example-sharded batches (any user on any rank, as the reference's loader deals examples,
        # xfmr_rec/data/lightning.py:109) through the fused user + item exchange; MF_BENCH_USER_MODE=partitioned times the
        # user-partitioned stream instead (user rows never travel)
        user_mode = os.environ.get("MF_BENCH_USER_MODE", "routed")
        def make_trainer():
            return mf.distributed.ShardedTrainer(mf, device, args.optimizer, args.num_negatives, num_users=NUM_USERS,
                                                 num_items=NUM_ITEMS, dim=DIM, logq=logq_table(device), user_mode=user_mode)
        comm_note = os.environ.get("MF_BENCH_COMM_NOTE")       # set when this process IS the restarted one
        try:
            trainer = make_trainer()
        except mf._lib.MfHipError as e:
            # default_comm raises on EVERY rank together when the C-side RCCL communicator fails its self-test (it has only
            # ever run on one rank where this was built).  RCCL is then in an unknown state HERE: fresh processes take over
            if os.environ.get("MF_COMM") in ("torch", "rccl"):
                raise
            restart_on_torch_collectives(f"mf_comm self-test failed: {e}")
        if user_mode == "partitioned":
            span_u = trainer.user_hi - trainer.user_lo
            for b in batches:
                b["user"] = trainer.user_lo + b["user"] % span_u
        # the sharded step prefetches the exchange plan of the batch after it (ids are known ahead)
        run_step = lambda j: trainer.step(batches[j % n_batches], next_b=batches[(j + 1) % n_batches])  # noqa: E731
        # everything that would stall the first sharded step happens before the spin-up: RCCL's lazy
        # initialisation (first collective) and the first batch's exchange plan (a host sync)
        warm = torch.zeros(world, dtype=torch.int64, device=device)
        torch.distributed.all_to_all_single(torch.empty_like(warm), warm)
        trainer.prefetch(batches[0])
        torch.cuda.synchronize()
        spin_up(mf, device, "train")
        for i in range(W):
            run_step(i)
        lib.mf_timing_reset()
        lib.mf_timing_enable(time_every(K * REPS))
        train_reps = [timed(lambda i, r=r: run_step(W + r * K + i), K, dist_on) / K * 1e3 for r in range(REPS)]
        dt_train = sorted(train_reps)[len(train_reps) // 2] * K / 1e3
        lib.mf_timing_enable(0)
        trainer.finish()                        # deferred error flags (id range / exchange capacity) of the timed steps
        spans = {n: kernel_span(lib, n)[0] for n in TRAIN_KERNELS}
        spans = {k: v for k, v in spans.items() if v}
    else:
        # cold first: what a step costs right after the device has idled (no spin-up; DESIGN.md 4, "post-idle ramp")
        time.sleep(0.05)
        cold = run_train_leg(mf, lib, device, batch=B, steps=min(K, 20), warmup=min(W, 3), optimizer=args.optimizer,
                             num_negatives=args.num_negatives, spin=False)
        cold_ms = cold["ms_per_step"]
        del cold
        # the reference's own inputs when the box has them (there is no network here, so normally it has not): the ids of a
        # MovieLens ratings file through the device batch producer, CSR positives, table heights from the data
        ml_batches, data_tag, ml_meta = None, "synthetic", None
        ml_file = mf.data.find_movielens()
        if ml_file is not None and os.environ.get("MF_BENCH_SYNTHETIC") != "1":
            table, ml_meta = mf.data.movielens_interactions(ml_file)
            sampler = table.sampler(num_items=ml_meta["num_items"], batch_size=B, seed=0, device=device)
            ml_batches = [flat_batch(sampler.batch(j)) for j in range(8)]
            data_tag = "movielens"
        leg = run_train_leg(mf, lib, device, batch=B, steps=K, warmup=W, optimizer=args.optimizer,
                            num_negatives=args.num_negatives, spin=True, reps=REPS, batches=ml_batches,
                            **({} if ml_meta is None else {"num_users": ml_meta["num_users"], "num_items": ml_meta["num_items"],
                                                           "use_logq": False}))
        dt_train, spans, trainer, batches = leg["ms_per_step"] * K / 1e3, leg["spans"], leg["trainer"], leg["batches"]
        train_reps = leg["reps_ms"]
    pairs_per_s = world * B * K / dt_train
    N = 2 * B
    train_roof = train_roofline(spans, B, DIM, world, args.optimizer, update_launches=2 if dist_on else 1)

    # ----------------------------------------------------------------- retrieval leg --
    n_users_leg = ml_meta["num_users"] if ml_meta else NUM_USERS
    n_items_leg = ml_meta["num_items"] if ml_meta else NUM_ITEMS
    with torch.no_grad():
        items = trainer.item_matrix()
        qrows = torch.arange(1 + rank * Q, 1 + (rank + 1) * Q, device=device) % n_users_leg
        queries = trainer.user_vectors(qrows)
    g = torch.Generator().manual_seed(7 + rank)
    item_w = zipf_weights(n_items_leg - 1)
    pieces, offs = [], [0]                         # per-query history (sorted unique item rows)
    for n in torch.randint(20, 300, (Q,), generator=g).tolist():
        pieces.append(torch.unique(torch.multinomial(item_w, n, replacement=True, generator=g) + 1))
        offs.append(offs[-1] + pieces[-1].numel())
    csr = (torch.tensor(offs, dtype=torch.int64, device=device), torch.cat(pieces).to(device))
    index = None
    if dist_on:
        searcher = mf.distributed.ShardedIndex(trainer.item_shard(), trainer.item_offset(), n_items_leg, stride=trainer.item_stride(),
                                               comm=trainer.comm)
        run_topk = lambda i: searcher.search(queries, TOP_K, exclude_csr=csr)   # noqa: E731
    else:
        index = mf.retrieval.ItemIndex(items)
        run_topk = lambda i: index.search(queries, TOP_K, exclude_csr=csr)      # noqa: E731
    spin_up(mf, device, "topk", index=mf.retrieval.ItemIndex(trainer.item_shard()) if dist_on else index)
    for i in range(W):
        run_topk(i)
    lib.mf_timing_reset()
    lib.mf_timing_enable(time_every(K * REPS))
    topk_reps = [timed(run_topk, K, dist_on) / K * 1e3 for _ in range(REPS)]
    dt_topk = sorted(topk_reps)[len(topk_reps) // 2] * K / 1e3
    lib.mf_timing_enable(0)
    qps = world * Q * K / dt_topk
    span, _n = kernel_span(lib, "topk_select")
    span_bf, _nb = kernel_span(lib, "topk_bf3")
    topk_roof = None
    # per launch every rank scores (world * Q) queries against its N / world rows
    n_local = items.shape[0] if not dist_on else trainer.item_shard().shape[0]
    flops = 2.0 * (world * Q) * n_local * DIM
    if span_bf:
        # the many-query path (mf_topk_bf3): a bf16 MFMA scan of every second 4-tile block of a bf16 copy of the catalog (the
        # seed: 1/2 of the algorithmic flops) and a full one (1 x), each tile staged through LDS ONCE per pass for the
        # <= 1024 queries of a workgroup, then exact fp32 rescoring of ~40 rows per query.  `achieved` counts the ALGORITHMIC
        # 2 Q N d only; one event pair spans the five launches (prep, seed, bound, scan, final).
        ach = flops / (span_bf * 1e-3) / 1e12
        qblocks = (world * Q + 1023) // 1024
        staged = 1.5 * qblocks * n_local * DIM * 2
        topk_roof = {"kernel": "bf3_prep_kernel + bf3_scan_kernel x2 (seed: half the catalog; full) + bf3_bound_kernel + bf3_final_kernel",
                     "bound": "mfma", "achieved": round(ach, 2),
                     "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4),
                     "traffic": measured_traffic("topk_bf3") if (Q, DIM, world) == (1024, 128, 1) else None,
                     "avg_ms": round(span_bf, 4), "launches": 5, "mfma_dtype": "bf16 (fp32 accumulate), exact fp32 rescoring",
                     "executed_mfma_frac": round(1.5 * ach / PEAK_BF16_MFMA_TFLOPS, 4),
                     "of_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                     "lds_dma": {"staged_GB_per_call": round(staged / 1e9, 4), "achieved_GBps": round(staged / (span_bf * 1e-3) / 1e9, 1),
                                 "chip_rate_GBps": LDS_DMA_CHIP_GBS, "frac": round(staged / (span_bf * 1e-3) / 1e9 / LDS_DMA_CHIP_GBS, 4)}}
    elif span:
        ach = flops / (span * 1e-3) / 1e12
        # one event pair spans the selection of a call: seeding pass over 1/8 of the catalog (which adds 1/8
        # to the flops actually issued; `achieved` counts the ALGORITHMIC 2 Q N d only), bound, main pass
        topk_roof = {"kernel": "select_seed_kernel + select_bound_kernel + select_kernel<RetrievalPolicy>", "bound": "mfma",
                     "achieved": round(ach, 2),
                     "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                     "traffic": measured_traffic("topk_select") if (Q, DIM, world) == (1024, 128, 1) else None,
                     "avg_ms": round(span, 4)}

    # ------------------------------------------------- SURVEY 8(d) matrix (N = 1 only) ----
    if not dist_on and not args.no_extras:
        def brief(leg, batch, dim, optimizer="adam"):
            roof = train_roofline(leg["spans"], batch, dim, 1, optimizer)
            return {"ms_per_step": round(leg["ms_per_step"], 4), "pairs_per_s": round(leg["pairs_per_s"], 1),
                    **({"reps_ms_per_step": spread(leg["reps_ms"])} if len(leg.get("reps_ms", ())) > 1 else {}),
                    "roofline": None if roof is None else {k: roof[k] for k in ("kernel", "achieved", "peak", "frac", "avg_ms", "of_fp32_mfma_peak", "executed_mfma_frac",
                                                                                "all_sweeps_frac", "hbm_kernels") if k in roof}}

        extras["q_small"] = topk_small_leg(mf, lib, index, device, DIM, rank)
        del index
        # the reference's DEFAULT training configuration (xfmr_rec/lightning.py:38-39): PairwiseHingeLoss, num_negatives = 4
        leg = run_train_leg(mf, lib, device, batch=B, steps=40, warmup=5, loss="PairwiseHingeLoss", num_negatives=4, use_logq=False, reps=3)
        extras["mined"] = {"workload": "C3 shape, PairwiseHingeLoss, num_negatives=4 (reference default loss), row-adam",
                           **brief(leg, B, DIM)}
        del leg
        # the same step captured in a hipGraph and replayed: with the search on the bf16 cores the eager step is bounded by the
        # host's ~22 launches and Python, not by the GPU (spans cannot be recorded inside a replayed graph: none here)
        leg = run_train_leg(mf, lib, device, batch=B, steps=40, warmup=5, loss="PairwiseHingeLoss", num_negatives=4, use_logq=False, reps=3,
                            graph=True, spin=False)
        extras["mined"]["graph_ms_per_step"] = round(leg["ms_per_step"], 4)
        extras["mined"]["graph_pairs_per_s"] = round(leg["pairs_per_s"], 1)
        del leg
        # config C2: MovieLens-1M shape, d = 64, in-batch sampled softmax
        leg = run_train_leg(mf, lib, device, batch=B, steps=40, warmup=5, num_users=6041, num_items=3884, dim=64, use_logq=False, reps=3)
        extras["c2_ml1m_d64"] = {"workload": "C2: MovieLens-1M shape (6,040 x 3,883), d=64, InfoNCE, row-adam", **brief(leg, B, 64)}
        del leg
        # positive lists of 1024 ids per user (SURVEY 8d stress)
        leg = run_train_leg(mf, lib, device, batch=B, steps=20, warmup=3, pos_pad=1024, reps=3)
        extras["pos_pad_1024"] = {"workload": "C3 shape, InfoNCE + logQ, P = 1024 padded positives per user", **brief(leg, B, DIM)}
        del leg
        # ALL of a user's positives, as the reference passes them (data/lightning.py:274-280), at MovieLens-25M's list-length
        # profile (log-normal, heaviest users >= 10^4 items, users drawn in proportion to their list length): the producer's
        # CSR lists read in place by the mask kernels -- no [B, P] tensor.  To hold against the P = 64 headline step.
        inter = make_csr_interactions(seed=0)
        sampler = mf.data.DeviceInteractionSampler(inter["pair_user"], inter["pair_item"], inter["pair_target"], inter["pos_off"],
                                                   inter["pos_items"], num_items=NUM_ITEMS, batch_size=B, seed=1, device=device)
        csr_batches = [flat_batch(sampler.batch(j)) for j in range(8)]
        per_batch = float(sum(int(inter["lens"].to(device)[b_["user"]].sum()) for b_ in csr_batches)) / len(csr_batches)
        leg = run_train_leg(mf, lib, device, batch=B, steps=40, warmup=5, batches=csr_batches, reps=3)
        extras["csr_positives_ml25m_lists"] = {
            "workload": "C3 shape, InfoNCE + logQ, ALL positives of every user as CSR lists (log-normal lengths, mean "
                        f"{float(inter['lens'].double().mean()):.0f}, longest {int(inter['lens'].max())}; {per_batch / 1e6:.2f} M list "
                        "entries looked up per batch), device batch producer", "reps_ms_per_step": spread(leg["reps_ms"]),
            "vs_headline_P64": round(leg["ms_per_step"] / (dt_train / K * 1e3), 4), **brief(leg, B, DIM)}
        del leg, sampler, csr_batches, inter
        # the same headline workload with the SGD update (36 d bytes / pair against HBM)
        leg = run_train_leg(mf, lib, device, batch=B, steps=40, warmup=5, optimizer="sgd", reps=3)
        extras["c3_sgd"] = {"workload": "C3 shape, InfoNCE + logQ, sparse SGD update", "reps_ms_per_step": spread(leg["reps_ms"]),
                            **brief(leg, B, DIM, "sgd")}
        del leg
        # config C5's width: d = 256, dense tables of the C3 shape, then hash / bloom towers over ONE GPU's share of C5
        # (100 M items x 10 M users / 8 ranks: 12.5 M + 1.25 M bucket rows x 256 = 12.8 + 1.3 GB of tables, x 3 with Adam's
        # moments; ids drawn from the full 10 M x 100 M id space, items log-uniform ~ Zipf(1))
        leg = run_train_leg(mf, lib, device, batch=B, steps=20, warmup=3, dim=256, reps=3)
        extras["c5_d256_dense"] = {"workload": "C3 table shape at C5's width d=256, InfoNCE + logQ, row-adam",
                                   "reps_ms_per_step": spread(leg["reps_ms"]), **brief(leg, B, 256)}
        del leg
        torch.cuda.empty_cache()
        hb = hashed_batches(8, B, device, users=10_000_000, items=100_000_000)
        leg = run_train_leg(mf, lib, device, batch=B, steps=20, warmup=3, dim=256, reps=3, batches=hb, num_users=1_250_000,
                            num_items=12_500_000, num_hashes=2, use_logq=False)
        extras["c5_d256_hashed"] = {"workload": "C5 per-GPU share: hash / bloom towers (2 hashes), 12.5 M item + 1.25 M user bucket "
                                                "rows x d=256 (12.8 + 1.3 GB tables + Adam moments), ids from 10 M users x 100 M "
                                                "items, InfoNCE, row-adam", "reps_ms_per_step": spread(leg["reps_ms"]),
                                    **brief(leg, B, 256)}
        # ... and the full-shard top-k of that catalog share: 1024 queries against 12.5 M rows x 256
        with torch.no_grad():
            shard = leg["trainer"].towers["item"].weight.detach()
            qv = torch.nn.functional.normalize(torch.randn(Q, 256, device=device), dim=-1)
            big = mf.retrieval.ItemIndex(shard)
            for _ in range(2):
                big.search(qv, TOP_K)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                big.search(qv, TOP_K)
            torch.cuda.synchronize()
            dt_big = (time.perf_counter() - t0) / 3
        extras["c5_shard_topk"] = {"workload": f"top-{TOP_K} of {Q} queries over 12.5 M rows x d=256 (one GPU's share of the 100 M-item catalog)",
                                   "ms_per_call": round(dt_big * 1e3, 3), "queries_per_s": round(Q / dt_big, 1),
                                   "algorithmic_TFLOPs": round(2.0 * Q * shard.shape[0] * 256 / dt_big / 1e12, 1)}
        del leg, big, shard, hb
        torch.cuda.empty_cache()
        # small batches: eager against one hipGraph replay per step
        ref_default = {"loss": "PairwiseHingeLoss", "num_negatives": 4, "use_logq": False}
        for name, bsz, kw in (("b1024", 1024, {}), ("b32", 32, {}),
                              # the reference's DEFAULT configuration: BATCH_SIZE = 32 (params.py:18), hidden_size = 32
                              # (lightning.py:33), PairwiseHingeLoss with 4 mined negatives (lightning.py:38-39); table heights of ML-25M
                              ("b32_reference_default", 32, {**ref_default, "dim": 32}),
                              # ... and the same step at the C3 table width (round 2 measured this leg at d = 128)
                              ("b32_reference_default_d128", 32, ref_default)):
            eager = run_train_leg(mf, lib, device, batch=bsz, steps=200, warmup=20, spin=False, **kw)
            graphed = run_train_leg(mf, lib, device, batch=bsz, steps=200, warmup=20, spin=False, graph=True, **kw)
            dim_leg = kw.get("dim", DIM)
            extras[name] = {"workload": f"ML-25M table heights, d = {dim_leg}, B = {bsz}, " + (kw.get("loss", "InfoNCE + logQ")) + ", row-adam",
                            "eager_ms_per_step": round(eager["ms_per_step"], 4), "graph_ms_per_step": round(graphed["ms_per_step"], 4),
                            "eager_pairs_per_s": round(eager["pairs_per_s"], 1), "graph_pairs_per_s": round(graphed["pairs_per_s"], 1)}
            del eager, graphed
            if kw.get("num_negatives"):
                # the same step in ONE launch (mf_step_small: bit-identical tables): device time per step (median of ten
                # 200-step regions, HIP events), wall time of the loop, phase stamps of the last step
                extras[name].update(small_step_leg(mf, device, bsz, {k: v for k, v in kw.items() if k != "dim"}, dim=dim_leg))
        torch.cuda.empty_cache()

    # --------------------------------------------------------------------- CPU leg ----
    cpu = None
    if rank == 0 and world == 1 and not dist_on and not args.no_cpu_baseline:
        if ml_meta is not None:      # (the CPU restatement takes padded positives: it runs the synthetic C3 batch in that case)
            batches, _ = make_batches(1, B, seed=1000, device=device)
        cb = [{k: v.cpu() for k, v in b.items()} for b in batches[:1]]
        cpu = cpu_baselines(args, cb, queries.cpu(), items.cpu(), pieces)

    if rank == 0:
        line = {
            "metric": "train pairs/sec + full-catalog top-k queries/sec, ML-25M d=128",
            "value": round(pairs_per_s, 1), "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt_train / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": data_tag,
            "reps_ms_per_step": spread(train_reps),      # REPS timed regions of K steps each; value / ms_per_step = the median one
            "config": {"workload": (f"MovieLens ratings file {ml_meta['path']} ({ml_meta['num_users'] - 1:,} users x "
                                    f"{ml_meta['num_items'] - 1:,} items, {ml_meta['train_pairs']:,} train pairs), d=128, InfoNCE, CSR positives, "
                                    if ml_meta else
                                    "C3: MovieLens-25M shape (162,541 users x 62,423 items), d=128, InfoNCE + logQ, ") +
                                   f"num_negatives={args.num_negatives}, row-{args.optimizer} update",
                       "batch_per_gpu": B, "items_per_step": N, "pos_pad": POS_PAD,
                       "parallelism": f"dp{world}" + (f" + both tables row-sharded, {trainer.user_mode} users" if dist_on else "")},
            "rccl_ranks": trainer.comm.rccl_ranks if dist_on else None,
            "transport": trainer.comm.transport if dist_on else "none (one process, one GPU)",
            **({"comm_note": comm_note} if dist_on and comm_note else {}),
            "cold_ms_per_step": None if cold_ms is None else round(cold_ms, 4),
            "roofline": train_roof,
            "topk": {"value": round(qps, 1), "unit": "queries/s", "ms_per_step": round(dt_topk / K * 1e3, 4),
                     "reps_ms_per_step": spread(topk_reps), "queries_per_gpu": Q, "k": TOP_K, "catalog_rows": n_items_leg,
                     "roofline": topk_roof},
            "extras": extras,
            "cpu_baseline": cpu,
        }
        # the second half of the headline metric and the other figures with a bar attached, as TOP-LEVEL scalars at the END
        # of the line: the driver's record keeps top-level scalars and the last 2 kB
        def dig(obj, *path):
            for key in path:
                obj = obj.get(key) if isinstance(obj, dict) else None
            return obj
        ref_small = dig(extras, "b32_reference_default", "one_launch_device_ms_per_step")
        ref_small128 = dig(extras, "b32_reference_default_d128", "one_launch_device_ms_per_step")
        avg = dig(train_roof, "all_kernels_avg_ms") or {}
        sweeps_ms = sum(avg.get(k, 0.0) for k in ("loss_fwd_dense", "loss_bwd_du", "loss_bwd_dv"))
        line.update({
            "topk_value": round(qps, 1), "topk_unit": "queries/s", "topk_ms_per_step": round(dt_topk / K * 1e3, 4),
            "topk_frac": dig(topk_roof, "frac"), "topk_device_ms": dig(topk_roof, "avg_ms"), "topk_traffic_bytes": dig(topk_roof, "traffic"),
            "q1_latency_us": dig(extras, "q_small", "q1", "latency_us"), "q32_latency_us": dig(extras, "q_small", "q32", "latency_us"),
            "one_launch_us": None if ref_small is None else round(ref_small * 1e3, 2),
            "one_launch_d128_us": None if ref_small128 is None else round(ref_small128 * 1e3, 2),
            "train_frac": dig(train_roof, "frac"), "train_tail_us": round((dt_train / K - sweeps_ms * 1e-3) * 1e6, 1) if sweeps_ms else None,
            "mined_frac": dig(extras, "mined", "roofline", "frac"), "mined_of_fp32_peak": dig(extras, "mined", "roofline", "of_fp32_mfma_peak"),
            "mined_ms_per_step": dig(extras, "mined", "ms_per_step"), "mined_pairs_per_s": dig(extras, "mined", "pairs_per_s"),
            "mined_graph_ms_per_step": dig(extras, "mined", "graph_ms_per_step"),
            "c2_ms_per_step": dig(extras, "c2_ml1m_d64", "ms_per_step"),
            "cpu_pairs_per_s": dig(cpu, "value"), "cpu_topk_queries_per_s": dig(cpu, "topk_value"),
        })
        print(json.dumps(line), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
