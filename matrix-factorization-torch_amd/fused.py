"""The reference's default training step in ONE launch (``mf_step_small``).

The reference trains with ``BATCH_SIZE = 32`` pairs (xfmr_rec/params.py:18), ``PairwiseHingeLoss`` and 4 mined negatives
(xfmr_rec/lightning.py:38-39; ``training_step`` :189-192, ``compute_losses`` :97-147, ``configure_optimizers`` :238-239).
At that size the multi-kernel path -- tower gathers, loss forward, backward, two sparse updates: 11+ launches -- is
bound by launch latency.  :class:`FusedSmallStep` runs the same step (same kernels' arithmetic, bit for bit: tests/
test_gpu_module.py compares the tables after several steps) in one workgroup and one launch for B <= 128 pairs and a
mined loss; anything else falls back to the ordinary path, step by step, with identical results.

    step = FusedSmallStep(towers, optimizer, loss_fn)        # EmbeddingTower tables, SparseSGD / RowAdam, any loss class
    loss = step(batch)                                        # batch: user [B], item [2B], target [B], pos [B, P] or pos_csr
"""
from __future__ import annotations

import torch

from . import _lib, losses, models, optim

MAX_BATCH, MAX_ITEMS = 128, 256


class FusedSmallStep:
    def __init__(self, towers: torch.nn.ModuleDict, optimizer: torch.optim.Optimizer, loss_fn: losses.EmbeddingLoss, *,
                 logq_table: torch.Tensor | None = None, all_losses: bool = False) -> None:
        if not isinstance(towers["user"], models.EmbeddingTower) or not isinstance(towers["item"], models.EmbeddingTower):
            raise _lib.MfHipError("FusedSmallStep takes plain embedding-table towers (hash towers use the ordinary step)")
        if not isinstance(optimizer, (optim.SparseSGD, optim.RowAdam)):
            raise _lib.MfHipError("FusedSmallStep takes optim.SparseSGD or optim.RowAdam")
        self.towers, self.opt, self.loss_fn = towers, optimizer, loss_fn
        self.set_logq_table(logq_table)
        self.kind_mask = (1 << len(losses.KINDS)) - 1 if all_losses else 1 << loss_fn.kind
        self.user, self.item = towers["user"].weight, towers["item"].weight
        if self.user.shape[1] != self.item.shape[1] or towers["user"].normalize != towers["item"].normalize:
            raise _lib.MfHipError("both towers must have the same width and normalisation")
        self._ws: torch.Tensor | None = None
        self._one = None
        self.fused_steps = self.fallback_steps = 0
        if isinstance(optimizer, optim.RowAdam):
            optimizer.init_state()
        self._group = {id(p): g for g in optimizer.param_groups for p in g["params"]}

    def set_logq_table(self, logq_table: torch.Tensor | None) -> None:
        """The logQ table the NEXT steps use (None: no correction) -- read at every call, so a table set or refreshed after
        the first step is honoured (the module's ``logq`` may appear after the first fused step)."""
        self.logq_table = None if logq_table is None else _lib.dev_f32(logq_table, "logq_table").reshape(-1)

    def supported(self, batch) -> bool:
        b, n = batch["user"].numel(), batch["item"].numel()
        k = int(self.loss_fn.num_negatives)
        gu, gi = self._group[id(self.user)], self._group[id(self.item)]
        same = all(gu[key] == gi[key] for key in gu if key != "params")
        return (b <= MAX_BATCH and b <= n <= MAX_ITEMS and 0 < k < n and k <= losses.MAX_MINED_NEGATIVES and same
                and self.loss_fn.kind != 0 and not getattr(self.opt, "capturable", False))

    def _fallback(self, b) -> torch.Tensor:
        self.fallback_steps += 1
        u = self.towers["user"](b["user"])
        v = self.towers["item"](b["item"])
        loss = self.loss_fn(u, v, b["target"], item_idx=b["item"], pos_idx=b.get("pos"), logq_table=self.logq_table, pos_csr=b.get("pos_csr"))
        if self._one is None:
            self._one = torch.ones((), device=loss.device)
        loss.backward(self._one)
        self.opt.step()
        return loss.detach()

    @torch.no_grad()
    def __call__(self, batch) -> torch.Tensor:
        """One training step on ``batch`` (flat dict: ``user`` [B], ``item`` [N = 2B: positives then negatives], ``target``
        [B], ``pos`` [B, P] or ``pos_csr``); returns the trained loss (detached: its backward has already been applied)."""
        if not self.supported(batch):
            with torch.enable_grad():
                return self._fallback(batch)
        lib = _lib.lib()
        user_ids, item_ids = _lib.dev_i64(batch["user"], "user"), _lib.dev_i64(batch["item"], "item")
        tgt = batch["target"]
        t = tgt.contiguous() if (tgt.is_cuda and tgt.dtype == torch.int64) else _lib.dev_f32(tgt, "target")
        pos, csr = batch.get("pos"), batch.get("pos_csr")
        pi, p = None, 0
        if csr is not None:
            _uid, off, items, n_pos_users = losses._csr_args((user_ids, csr[1], csr[2]), user_ids.numel())
        elif pos is not None and pos.shape[1] > 0:
            pi = _lib.dev_i64(pos, "pos")
            p = pi.shape[1]
        d = self.user.shape[1]
        if self._ws is None:
            self._ws = _lib.workspace(lib.mf_step_small_ws_bytes(d), self.user.device)
        out = torch.empty(len(losses.KINDS), dtype=torch.float32, device=self.user.device)
        group = self._group[id(self.user)]
        adam = isinstance(self.opt, optim.RowAdam)
        if adam:
            su, si = self.opt.state[self.user], self.opt.state[self.item]
            if su["step"] != si["step"]:
                raise _lib.MfHipError("the two tables' Adam step counts differ: they were not always stepped together")
            b1, b2 = group["betas"]
            # the counters are committed only after the launch has been accepted (a refused call must not advance the
            # bias corrections of later steps -- ADVICE r3)
            args_opt = (1, su["step"] + 1, None, group["lr"], b1, b2, group["eps"], group["weight_decay"])
            state = (su["exp_avg"].data_ptr(), su["exp_avg_sq"].data_ptr(), si["exp_avg"].data_ptr(), si["exp_avg_sq"].data_ptr())
        else:
            args_opt = (0, 1, None, group["lr"], 0.0, 0.0, 0.0, group["weight_decay"])
            state = (None, None, None, None)
        fn = self.loss_fn
        lq = self.logq_table
        _lib.check(lib.mf_step_small(
            self.user.data_ptr(), state[0], state[1], self.user.shape[0], self.item.data_ptr(), state[2], state[3], self.item.shape[0],
            d, int(self.towers["user"].normalize), user_ids.data_ptr(), item_ids.data_ptr(), t.data_ptr(), int(t.dtype == torch.int64),
            _lib.ptr(pi), p, None if csr is None else off.data_ptr(), None if csr is None else items.data_ptr(),
            0 if csr is None else n_pos_users, user_ids.numel(), item_ids.numel(), fn.kind, self.kind_mask, int(fn.num_negatives),
            float(fn.sigma), float(fn.margin), _lib.ptr(lq), 0 if lq is None else lq.numel(), *args_opt, self._ws.data_ptr(),
            self._ws.numel(), out.data_ptr(), _lib.stream_ptr()))
        if adam:
            su["step"] += 1
            si["step"] += 1
        self.fused_steps += 1
        self.losses = out
        return out[fn.kind]
