"""Sparse row optimisers for :class:`models.EmbeddingTower` tables.

``configure_optimizers`` of the reference returns dense ``torch.optim.AdamW``
(xfmr_rec/lightning.py:238-239).  With embedding tables only the gathered rows
receive gradient, so the update is a scatter over the touched rows
(``mf_update_sgd`` / ``mf_update_adam``): duplicate ids are summed first, every
touched row is read and written once.  Spec: ``oracle/embed.py`` (no reference
implementation).  Both classes are ``torch.optim.Optimizer``
subclasses, so Lightning's automatic optimisation can drive them.
"""
from __future__ import annotations

import os

import torch

from . import _lib


def _pending(p: torch.Tensor):
    items = getattr(p, "_mf_pending", None)
    if not items:
        return None
    if len(items) == 1:
        ids, g, norm = items[0]
    else:
        if len({n for _, _, n in items}) != 1:
            raise _lib.MfHipError("a table was gathered both with and without normalisation in one step")
        ids = torch.cat([i for i, _, _ in items])
        g = torch.cat([x for _, x, _ in items])
        norm = items[0][2]
    return ids, g, norm


_side_streams: dict = {}


def run_table_jobs(jobs) -> None:
    """Run independent per-table update jobs (callables issuing work on the *current* stream).  Default: one
    after the other on the caller's stream.  With ``MF_TABLE_STREAMS=1`` the first runs on the caller's stream
    and the others on cached side streams, forked from and joined back with events: the chains do overlap
    (~35 us of small kernels), but on this part a cross-stream join costs a 20-30 us bubble in the stream it
    rejoins -- measured at B = 8192: 1.1767 ms / step with two streams against 1.1644 on one."""
    if not jobs:
        return
    if len(jobs) == 1 or os.environ.get("MF_TABLE_STREAMS") != "1":
        for job in jobs:
            job()
        return
    cur = torch.cuda.current_stream()
    fork = torch.cuda.Event()
    fork.record(cur)
    joins = []
    for k, job in enumerate(jobs[1:]):
        key = (cur.device, k)
        side = _side_streams.get(key)
        if side is None:
            side = _side_streams[key] = torch.cuda.Stream(device=cur.device)
        side.wait_event(fork)
        with torch.cuda.stream(side):
            job()
            done = torch.cuda.Event()
            done.record(side)
        joins.append(done)
    jobs[0]()
    for done in joins:
        cur.wait_event(done)


def _pair_update(adam: bool, jobs, step: int, step_dev, hyper: dict) -> bool:
    """Both tables of the step in ONE launch (``mf_update_pair``) when there are exactly two pending tables of one width and one
    parameter group -- the usual step: user rows and item rows.  ``jobs``: (table, state or None, ids, grad, normalized).
    False: not that shape (or lists beyond the one-launch range): the caller updates the tables one by one."""
    if len(jobs) != 2 or os.environ.get("MF_UPDATE_PAIR", "1") != "1":
        return False
    (pa, sa, ia, ga, na), (pb, sb, ib, gb, nb) = jobs
    d = pa.shape[1]
    if pb.shape[1] != d or not (0 < ia.numel() <= 65536 and 0 < ib.numel() <= 65536):
        return False
    lib = _lib.lib()
    wa = _lib.workspace(lib.mf_update_ws_bytes(ia.numel(), d), pa.device)
    wb = _lib.workspace(lib.mf_update_ws_bytes(ib.numel(), d), pb.device)
    b1, b2 = hyper.get("betas", (0.0, 0.0))
    _lib.check(lib.mf_update_pair(
        int(adam), d,
        pa.data_ptr(), sa["exp_avg"].data_ptr() if adam else None, sa["exp_avg_sq"].data_ptr() if adam else None, pa.shape[0],
        ia.data_ptr(), ia.numel(), ga.data_ptr(), int(na), wa.data_ptr(), wa.numel(),
        pb.data_ptr(), sb["exp_avg"].data_ptr() if adam else None, sb["exp_avg_sq"].data_ptr() if adam else None, pb.shape[0],
        ib.data_ptr(), ib.numel(), gb.data_ptr(), int(nb), wb.data_ptr(), wb.numel(),
        step, step_dev, hyper["lr"], b1, b2, hyper.get("eps", 0.0), hyper["weight_decay"], _lib.stream_ptr()))
    return True


class _SparseRowOptimizer(torch.optim.Optimizer):
    def zero_grad(self, set_to_none: bool = True) -> None:
        super().zero_grad(set_to_none=set_to_none)
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_mf_pending", None):
                    p._mf_pending.clear()

    @staticmethod
    def _check(p: torch.Tensor) -> None:
        if p.grad is not None:
            raise _lib.MfHipError("dense gradient on an embedding table: gather rows through EmbeddingTower")
        if p.dim() != 2 or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            raise _lib.MfHipError("embedding table must be a contiguous 2-D fp32 tensor on the GPU")


class SparseSGD(_SparseRowOptimizer):
    """row -= lr * (sum of the row's gradients + weight_decay * row), touched rows only."""

    def __init__(self, params, lr: float = 1e-2, weight_decay: float = 0.0) -> None:
        super().__init__(params, {"lr": lr, "weight_decay": weight_decay})

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        jobs, done, pairs = [], [], []
        for group in self.param_groups:
            for p in group["params"]:
                self._check(p)
                pend = _pending(p)
                if pend is None:
                    continue
                pairs.append((p, None, pend[0], pend[1], pend[2], group))

                def job(p=p, pend=pend, group=group):
                    ids, g, norm = pend
                    n, d = ids.numel(), p.shape[1]
                    ws = _lib.workspace(lib.mf_update_ws_bytes(n, d), p.device)
                    _lib.check(lib.mf_update_sgd(p.data_ptr(), p.shape[0], d, ids.data_ptr(), n, g.data_ptr(), int(norm),
                                                 group["lr"], group["weight_decay"], ws.data_ptr(), ws.numel(),
                                                 _lib.stream_ptr()))

                jobs.append(job)
                done.append(p)
        one_group = len({id(x[5]) for x in pairs}) == 1
        if not (one_group and _pair_update(False, [x[:5] for x in pairs], 1, None, pairs[0][5] if pairs else {})):
            run_table_jobs(jobs)
        for p in done:
            p._mf_pending.clear()
        return loss


class RowAdam(_SparseRowOptimizer):
    """Lazy row-wise AdamW: moments of touched rows only, global step for the bias
    correction, decoupled weight decay (torch.optim.AdamW's defaults otherwise)."""

    def __init__(self, params, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01) -> None:
        super().__init__(params, {"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay})
        # capturable: the global step lives in device memory and is bumped by a kernel of the step itself, so that a
        # step captured in a hipGraph (``graph.CapturedStep``) keeps counting when it is replayed -- a by-value
        # argument would be frozen at capture.  Same bits either way: the bias correction is evaluated on the device.
        self.capturable = False

    def on_replay(self) -> None:
        """Called once per graph replay of a captured step: keeps the host-side step counts in line."""
        for group in self.param_groups:
            for p in group["params"]:
                if self.state[p] and "step_t" in self.state[p]:
                    self.state[p]["step"] += 1

    def init_state(self) -> None:
        """Allocate the moment tables now instead of inside the first ``step`` (two table-sized allocations and
        memsets per table: a multi-millisecond stall in the middle of an otherwise steady first step)."""
        for group in self.param_groups:
            for p in group["params"]:
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p)
                    state["exp_avg_sq"] = torch.zeros_like(p)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        jobs, done, pairs = [], [], []
        for group in self.param_groups:
            for p in group["params"]:
                self._check(p)
                pend = _pending(p)
                if pend is None:
                    continue
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p)
                    state["exp_avg_sq"] = torch.zeros_like(p)
                step_dev = None
                if self.capturable:
                    if "step_t" not in state:
                        state["step_t"] = torch.full((1,), state["step"], dtype=torch.int64, device=p.device)
                    state["step_t"] += 1                   # a kernel: replayed with the graph
                    step_dev = state["step_t"].data_ptr()
                if not (self.capturable and torch.cuda.is_current_stream_capturing()):
                    state["step"] += 1                     # (a capture runs no kernel: the replays count, via on_replay)
                pairs.append((p, state, pend[0], pend[1], pend[2], group, step_dev))

                def job(p=p, pend=pend, group=group, state=state, step_dev=step_dev):
                    ids, g, norm = pend
                    n, d = ids.numel(), p.shape[1]
                    ws = _lib.workspace(lib.mf_update_ws_bytes(n, d), p.device)
                    b1, b2 = group["betas"]
                    _lib.check(lib.mf_update_adam(p.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                                  p.shape[0], d, ids.data_ptr(), n, g.data_ptr(), int(norm), state["step"],
                                                  step_dev, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                                  ws.data_ptr(), ws.numel(), _lib.stream_ptr()))

                jobs.append(job)
                done.append(p)
        # the usual step -- user rows and item rows, one parameter group, the same global step -- is ONE launch
        same = (len(pairs) == 2 and pairs[0][5] is pairs[1][5] and pairs[0][1]["step"] == pairs[1][1]["step"]
                and (pairs[0][6] is None) == (pairs[1][6] is None))
        if not (same and _pair_update(True, [x[:5] for x in pairs], pairs[0][1]["step"], pairs[0][6], pairs[0][5])):
            run_table_jobs(jobs)
        for p in done:
            p._mf_pending.clear()
        return loss
