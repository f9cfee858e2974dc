"""The seven in-batch embedding losses of ``xfmr_rec/losses.py`` on gfx950.

Same class names, constructor keywords (``num_negatives``, ``sigma``, ``margin``;
losses.py:27-37), call signature (``loss_fn(user_embed, item_embed, target, *,
item_idx, pos_idx)``; losses.py:39-52) and ``ValueError`` behaviour
(``check_inputs``, losses.py:54-79) as the reference, so ``get_loss_fns``
(xfmr_rec/lightning.py:267-287) can instantiate them unchanged.  The arithmetic runs
in ``libmf_hip.so`` (``mf_loss_fwd`` / ``mf_loss_bwd``); there is no torch fallback.

Extensions (absent upstream, default off): ``logq=`` (logQ correction
``L_ij -= logq[j]``), :func:`fused_losses` (all seven values from ONE pass over
the score tiles, as ``compute_losses`` evaluates all seven every step,
xfmr_rec/lightning.py:137-146) and ``pos_csr=(user_ids, pos_off, pos_items)``: the users' positive lists as the batch
producer keeps them in HBM (CSR over ALL users) instead of the padded ``pos_idx[B, P]`` -- the reference pads every batch
to its longest list (xfmr_rec/data/lightning.py:274-280), > 10^4 columns for MovieLens-25M; the masks are the same bits.
"""
from __future__ import annotations

import abc

import torch

from . import _lib

KINDS = (
    "AlignmentLoss",
    "ContrastiveLoss",
    "AlignmentContrastiveLoss",
    "InfomationNoiseContrastiveEstimationLoss",
    "MutualInformationNeuralEstimationLoss",
    "PairwiseHingeLoss",
    "PairwiseLogisticLoss",
)
MAX_MINED_NEGATIVES = 64


class _SquaredDistance(torch.autograd.Function):
    """``D[i, j] = 0.5 * max(|q_i|^2 + |c_j|^2 - 2 q_i . c_j, 0)`` with the dot products from the fp32 MFMA tile engine
    (``mf_scores``: the canonical chain of include/mf_numerics.h) and chain-ordered norms (``mf_row_sqnorm``)."""

    @staticmethod
    def forward(ctx, query_embed, candidate_embed):
        q = _lib.dev_f32(query_embed, "query_embed")
        c = _lib.dev_f32(candidate_embed, "candidate_embed")
        if q.dim() != 2 or c.dim() != 2 or q.shape[1] != c.shape[1]:  # noqa: PLR2004
            msg = f"expected (Q, d) and (N, d): {tuple(q.shape) = }, {tuple(c.shape) = }"
            raise ValueError(msg)
        d = q.shape[1]
        dp = _lib.padded_width(d)
        qp, cp = (q, c) if dp == d else (torch.nn.functional.pad(q, (0, dp - d)), torch.nn.functional.pad(c, (0, dp - d)))
        lib = _lib.lib()
        dot = torch.empty(q.shape[0], c.shape[0], dtype=torch.float32, device=q.device)
        nq = torch.empty(q.shape[0], dtype=torch.float32, device=q.device)
        nc = torch.empty(c.shape[0], dtype=torch.float32, device=q.device)
        st = _lib.stream_ptr()
        _lib.check(lib.mf_scores(qp.data_ptr(), qp.shape[0], cp.data_ptr(), cp.shape[0], dp, dot.data_ptr(), st))
        _lib.check(lib.mf_row_sqnorm(qp.data_ptr(), qp.shape[0], dp, nq.data_ptr(), st))
        _lib.check(lib.mf_row_sqnorm(cp.data_ptr(), cp.shape[0], dp, nc.data_ptr(), st))
        sq = torch.addcmul(nq[:, None] + nc[None, :], dot, dot.new_tensor(-2.0))      # mf_half_sqdist, elementwise
        ctx.save_for_backward(q, c, sq)
        ctx.dtypes = (query_embed.dtype, candidate_embed.dtype)
        return sq.clamp_min(0.0) * 0.5

    @staticmethod
    def backward(ctx, grad):
        q, c, sq = ctx.saved_tensors
        g = torch.where(sq > 0, grad.to(torch.float32), torch.zeros((), device=grad.device))   # the clamp's subgradient
        dq = g.sum(dim=1, keepdim=True) * q - g @ c              # d D_ij / d q_i = q_i - c_j
        dc = g.sum(dim=0)[:, None] * c - g.t() @ q
        return dq.to(ctx.dtypes[0]), dc.to(ctx.dtypes[1])


def squared_distance(query_embed: torch.Tensor, candidate_embed: torch.Tensor) -> torch.Tensor:
    """``cdist(q, c) ** 2 / 2`` (losses.py:9-12) as a ``[Q, N]`` matrix -- the public helper; the losses themselves
    never materialise it.  Differentiable (the backward is plain torch on the GPU: not the hot path)."""
    return _SquaredDistance.apply(query_embed, candidate_embed)


def weighted_mean(values: torch.Tensor, sample_weights: torch.Tensor, *, dim: int | None = None,
                  keepdim: bool = False) -> torch.Tensor:
    """losses.py:15-23: ``sum(values * w / (sum(w) + 1e-10))`` along ``dim``."""
    denominator = sample_weights.sum(dim=dim, keepdim=True) + 1e-10
    return (values * sample_weights / denominator).sum(dim=dim, keepdim=keepdim)


def _csr_args(pos_csr, batch: int):
    """(user_ids [B], pos_off [U + 1], pos_items, U) as contiguous int64 GPU tensors, checked."""
    if len(pos_csr) != 3:  # noqa: PLR2004
        msg = "pos_csr should be (user_ids, pos_off, pos_items)"
        raise ValueError(msg)
    uid, off, items = (_lib.dev_i64(t, name) for t, name in zip(pos_csr, ("pos_csr user_ids", "pos_csr pos_off", "pos_csr pos_items")))
    if uid.numel() != batch or off.numel() < 2:  # noqa: PLR2004
        msg = f"pos_csr needs one user id per batch row and at least one list: {uid.numel() = }, {batch = }, {off.numel() = }"
        raise ValueError(msg)
    if items.numel() == 0:
        items = torch.zeros(1, dtype=torch.int64, device=uid.device)     # keeps the pointer valid
    return uid, off, items, off.numel() - 1


def _prepare(user_embed, item_embed, target, item_idx, pos_idx, logq, logq_table=None):
    u = _lib.dev_f32(user_embed, "user_embed")
    v = _lib.dev_f32(item_embed, "item_embed")
    d = u.shape[1]
    dp = _lib.padded_width(d)
    if dp != d:  # zero columns change no score (mf_numerics.h)
        u = torch.nn.functional.pad(u, (0, dp - d))
        v = torch.nn.functional.pad(v, (0, dp - d))
    # the reference's target is the int64 rating (data/lightning.py:72-76): taken as it is, no conversion kernel
    t = target.contiguous() if (target.is_cuda and target.dtype == torch.int64) else _lib.dev_f32(target, "target")
    ii = _lib.dev_i64(item_idx, "item_idx")
    if ii.numel() != v.shape[0]:
        msg = f"item_idx should have one id per item row: {ii.numel() = }, {v.shape[0] = }"
        raise ValueError(msg)
    pi = None
    p = 0
    if pos_idx is not None:
        pi = _lib.dev_i64(pos_idx, "pos_idx")
        if pi.dim() != 2 or pi.shape[0] != u.shape[0]:
            msg = f"pos_idx should be (batch_size, num_positives): {tuple(pi.shape) = }"
            raise ValueError(msg)
        p = pi.shape[1]
        if p == 0:
            pi = None
    lq = None
    lq_rows = 0
    if logq is not None and logq_table is not None:
        msg = "pass either logq (one value per item row of the batch) or logq_table (looked up by item_idx), not both"
        raise ValueError(msg)
    if logq is not None:
        lq = _lib.dev_f32(logq, "logq")
        if lq.numel() != v.shape[0]:
            msg = f"logq should have one value per item row: {lq.numel() = }, {v.shape[0] = }"
            raise ValueError(msg)
    elif logq_table is not None:
        lq = _lib.dev_f32(logq_table, "logq_table").reshape(-1)
        lq_rows = lq.numel()
    return u, v, t, ii, pi, p, (lq, lq_rows), d, dp


class PreparedMasks:
    """Hit masks of one batch, built ahead of its forward (``EmbeddingLoss.prepare_masks``): they depend on
    the ids only, so they can be computed on a side stream beside the tower gathers.  Holds the loss
    workspace the forward will use and the event that orders the two streams."""

    def __init__(self, lease, event, key) -> None:
        self.lease, self.event, self.key = lease, event, key


_side_stream: dict = {}


def prepare_masks(item_idx, pos_idx, *, batch_size: int, embedding_dim: int, num_negatives: int = 0, pos_csr=None) -> PreparedMasks:
    ii = _lib.dev_i64(item_idx, "item_idx")
    pi = None if pos_idx is None or pos_idx.shape[1] == 0 else _lib.dev_i64(pos_idx, "pos_idx")
    p = 0 if pi is None else pi.shape[1]
    b, n, dp = int(batch_size), ii.numel(), _lib.padded_width(int(embedding_dim))
    csr = None if pos_csr is None else _csr_args(pos_csr, b)
    if csr is not None and pi is not None:
        msg = "pass either pos_idx (padded) or pos_csr, not both"
        raise ValueError(msg)
    lib = _lib.lib()
    cur = torch.cuda.current_stream()
    side = _side_stream.get(ii.device)
    if side is None:
        side = _side_stream[ii.device] = torch.cuda.Stream(device=ii.device, priority=-1)
    lease = _lib.LeasedWorkspace(lib.mf_loss_ws_bytes(b, n, dp, p, int(num_negatives)), ii.device)
    ws = lease.tensor
    side.wait_stream(cur)                       # the ids, and the workspace's previous use, are on the caller's stream
    with torch.cuda.stream(side):
        if csr is None:
            _lib.check(lib.mf_loss_masks(b, n, dp, p, int(num_negatives), _lib.ptr(ii), _lib.ptr(pi), _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr()))
        else:
            _lib.check(lib.mf_loss_masks_csr(b, n, dp, int(num_negatives), _lib.ptr(ii), _lib.ptr(csr[0]), _lib.ptr(csr[1]),
                                             _lib.ptr(csr[2]), csr[3], _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        event = torch.cuda.Event()
        event.record(side)
    for t in (ii,) + (() if pi is None else (pi,)) + (() if csr is None else csr[:3]):
        t.record_stream(side)
    pos_key = (None if pi is None else pi.data_ptr()) if csr is None else ("csr", csr[0].data_ptr(), csr[1].data_ptr())
    return PreparedMasks(lease, event, (b, n, dp, p, int(num_negatives), ii.data_ptr(), pos_key))


class _LossFunction(torch.autograd.Function):
    """One pass evaluates every loss in ``kind_mask``; backward differentiates ``bwd_kind``."""

    @staticmethod
    def forward(ctx, user_embed, item_embed, target, item_idx, pos_idx, logq, kind_mask, bwd_kind,
                num_negatives, sigma, margin, prepared=None, logq_table=None, train_kind=None, pos_csr=None):
        u, v, t, ii, pi, p, (lq, lq_rows), d, dp = _prepare(user_embed, item_embed, target, item_idx, pos_idx, logq, logq_table)
        b, n = u.shape[0], v.shape[0]
        csr = None if pos_csr is None else _csr_args(pos_csr, b)
        if csr is not None and pi is not None:
            msg = "pass either pos_idx (padded) or pos_csr, not both"
            raise ValueError(msg)
        lib = _lib.lib()
        if prepared is not None:
            pos_key = (None if pi is None else pi.data_ptr()) if csr is None else ("csr", csr[0].data_ptr(), csr[1].data_ptr())
            if prepared.key != (b, n, dp, p, int(num_negatives), ii.data_ptr(), pos_key):
                msg = "prepared masks belong to another batch / shape"
                raise ValueError(msg)
            torch.cuda.current_stream().wait_event(prepared.event)
            lease, ii_arg, ready = prepared.lease, ii, _lib.LOSS_MASKS_READY    # "the masks are in ws"
        else:
            lease, ii_arg, ready = _lib.LeasedWorkspace(lib.mf_loss_ws_bytes(b, n, dp, p, num_negatives), u.device), ii, 0
        ws = lease.tensor
        ctx.lease = lease                            # back to the pool when autograd drops this node
        out = torch.empty(len(KINDS), dtype=torch.float32, device=u.device)      # mf_loss_fwd writes all 7 entries
        flags = (_lib.LOSS_TARGET_I64 if t.dtype == torch.int64 else 0) | (_lib.LOSS_ROWC if bwd_kind is not None else 0) | ready
        if csr is None:
            _lib.check(lib.mf_loss_fwd(b, n, dp, p, num_negatives, sigma, margin, kind_mask, _lib.ptr(u), _lib.ptr(v),
                                       _lib.ptr(t), _lib.ptr(ii_arg), _lib.ptr(pi), _lib.ptr(lq), lq_rows, flags, _lib.ptr(ws),
                                       ws.numel(), _lib.ptr(out), None, _lib.stream_ptr()))
        else:
            _lib.check(lib.mf_loss_fwd_csr(b, n, dp, num_negatives, sigma, margin, kind_mask, _lib.ptr(u), _lib.ptr(v),
                                           _lib.ptr(t), _lib.ptr(ii_arg), _lib.ptr(csr[0]), _lib.ptr(csr[1]), _lib.ptr(csr[2]), csr[3],
                                           _lib.ptr(lq), lq_rows, flags, _lib.ptr(ws), ws.numel(), _lib.ptr(out), None,
                                           _lib.stream_ptr()))
        ctx.save_for_backward(u, v, ws)              # targets and logQ values live on in the workspace
        ctx.meta = (b, n, d, dp, p, num_negatives, sigma, margin, bwd_kind, user_embed.dtype, item_embed.dtype)
        ctx.train_kind = train_kind
        # one trained loss (``bwd_kind``): hand out the scalar itself, so that autograd needs no
        # select-backward (two fills and a copy per step) between ``loss.backward()`` and this node
        return out if bwd_kind is None else out[bwd_kind]

    @staticmethod
    def backward(ctx, grad_out):
        u, v, ws = ctx.saved_tensors
        b, n, d, dp, p, k, sigma, margin, bwd_kind, u_dtype, v_dtype = ctx.meta
        # only the train loss is back-propagated (xfmr_rec/lightning.py:192)
        # (fused call: the trained kind is named up front -- ``train_kind`` -- or found by a host-synchronising scan)
        if bwd_kind is not None:
            nz = [bwd_kind]
        elif ctx.train_kind is not None:
            nz = [ctx.train_kind]
        else:
            nz = torch.nonzero(grad_out).flatten().tolist()
        du = dv = None
        lib = _lib.lib()
        grad_out = grad_out.to(torch.float32).contiguous()          # no-ops for the usual fp32 gradient
        for kind in nz:
            g = grad_out.reshape(1) if bwd_kind is not None else grad_out[kind : kind + 1]   # views: no kernel
            du_k = torch.empty_like(u)
            dv_k = torch.empty_like(v)
            _lib.check(lib.mf_loss_bwd(b, n, dp, p, k, sigma, margin, kind, _lib.ptr(u), _lib.ptr(v),
                                       _lib.LOSS_ROWC if bwd_kind is not None else 0, _lib.ptr(ws), ws.numel(), _lib.ptr(g),
                                       _lib.ptr(du_k), _lib.ptr(dv_k), _lib.stream_ptr()))
            du = du_k if du is None else du + du_k
            dv = dv_k if dv is None else dv + dv_k
        if du is None:
            du, dv = torch.zeros_like(u), torch.zeros_like(v)
        if dp != d:
            du, dv = du[:, :d], dv[:, :d]
        return (du.to(u_dtype), dv.to(v_dtype)) + (None,) * 13


def check_inputs(user_embed: torch.Tensor, item_embed: torch.Tensor, target: torch.Tensor) -> None:
    """Same conditions and exception type as losses.py:54-79."""
    if user_embed.dim() != 2 or item_embed.dim() != 2:  # noqa: PLR2004
        msg = f"inputs should have 2 dimensions: {user_embed.dim() = }, {item_embed.dim() = }"
        raise ValueError(msg)
    if user_embed.size(1) != item_embed.size(1):
        msg = f"embeddings dimension 1 should match: {user_embed.size(1) = }, {item_embed.size(1) = }"
        raise ValueError(msg)
    if not (user_embed.size(0) == target.size(0) and item_embed.size(0) >= target.size(0)):
        msg = (
            "embeddings dimension 0 should match: "
            f"{target.size(0) = }, {user_embed.size(0) = }, {item_embed.size(0) = }"
        )
        raise ValueError(msg)


def _check_num_negatives(k: int, n_items: int) -> None:
    """Mining (0 < num_negatives < N, losses.py:137-141) keeps at most MAX_MINED_NEGATIVES per row on this path."""
    if 0 < k < n_items and k > MAX_MINED_NEGATIVES:
        msg = f"semi-hard mining supports num_negatives <= {MAX_MINED_NEGATIVES} (or >= num_items): {k = }"
        raise NotImplementedError(msg)


def fused_losses(user_embed, item_embed, target, *, item_idx, pos_idx, num_negatives=0, sigma=1.0, margin=1.0,
                 logq=None, logq_table=None, kinds=KINDS, train_loss: str | None = None, pos_csr=None) -> dict[str, torch.Tensor]:
    """All requested losses from one sweep over the score tiles.  Every returned value is
    differentiable; backward runs one HIP backward per loss that receives gradient.  ``train_loss`` names the one
    loss that will be back-propagated (xfmr_rec/lightning.py:192): the others are then constants of the backward,
    which no longer has to look (with a host sync) for the entries that received gradient."""
    check_inputs(user_embed, item_embed, target)
    _check_num_negatives(int(num_negatives), item_embed.size(0))
    mask = 0
    for name in kinds:
        mask |= 1 << KINDS.index(name)
    if train_loss is not None and train_loss not in kinds:
        msg = f"train_loss must be one of the evaluated kinds: {train_loss = }"
        raise ValueError(msg)
    out = _LossFunction.apply(user_embed, item_embed, target, item_idx, pos_idx, logq, mask, None,
                              int(num_negatives), float(sigma), float(margin), None, logq_table,
                              None if train_loss is None else KINDS.index(train_loss), pos_csr)
    return {name: out[KINDS.index(name)] for name in kinds}


class EmbeddingLoss(torch.nn.Module, abc.ABC):
    """Base class, mirrors ``EmbeddingLoss`` (losses.py:26-79)."""

    def __init__(self, *, num_negatives: int = 0, sigma: float = 1.0, margin: float = 1.0) -> None:
        super().__init__()
        self.num_negatives = num_negatives
        self.sigma = sigma
        self.margin = margin

    @property
    def kind(self) -> int:
        """Which of the seven kernels' losses this instance is.  Upstream a subclass is free to bring its own ``loss`` /
        ``score_loss_fn`` (losses.py:81-90, 342, 348-349) because everything is a torch expression; here the arithmetic is
        compiled into ``libmf_hip.so``, so a class (or a patched ``score_loss_fn``) the kernels do not know is refused
        loudly instead of being trained as something else."""
        known = next((c for c in type(self).__mro__ if c.__module__ == __name__ and c.__name__ in KINDS), None)
        if known is None:
            msg = (f"{type(self).__name__} is not one of the seven losses compiled into libmf_hip.so ({', '.join(KINDS)}): "
                   "a custom EmbeddingLoss needs its own kernel -- subclass one of the seven, or override loss() with torch code")
            raise NotImplementedError(msg)
        phi = known.__dict__.get("score_loss_fn")
        if phi is not None and ("score_loss_fn" in self.__dict__ or type(self).score_loss_fn is not phi):
            msg = (f"{type(self).__name__}.score_loss_fn differs from {known.__name__}'s: the kernels apply the built-in "
                   "softplus / relu per tile element and would ignore it")
            raise NotImplementedError(msg)
        return KINDS.index(known.__name__)

    def check_inputs(self, user_embed: torch.Tensor, item_embed: torch.Tensor, target: torch.Tensor) -> None:
        check_inputs(user_embed, item_embed, target)

    def forward(self, user_embed: torch.Tensor, item_embed: torch.Tensor, target: torch.Tensor, *,
                item_idx: torch.Tensor, pos_idx: torch.Tensor | None = None, logq: torch.Tensor | None = None,
                prepared: PreparedMasks | None = None, logq_table: torch.Tensor | None = None, pos_csr=None) -> torch.Tensor:
        """``logq`` (one value per item row of the batch) or ``logq_table`` (a table over all item rows, looked up by
        ``item_idx`` inside the kernel) switch on the logQ correction ``L_ij -= logq_j`` (our extension).
        ``pos_csr = (user_ids [B], pos_off [U + 1], pos_items)`` gives the positives as CSR lists over all users
        instead of the padded ``pos_idx`` (module docstring)."""
        self.check_inputs(user_embed, item_embed, target)
        return self.loss(user_embed, item_embed, target, item_idx=item_idx, pos_idx=pos_idx, logq=logq, prepared=prepared,
                         logq_table=logq_table, pos_csr=pos_csr)

    def prepare_masks(self, item_idx: torch.Tensor, pos_idx: torch.Tensor | None, *, batch_size: int,
                      embedding_dim: int, pos_csr=None) -> PreparedMasks:
        """Optional (no reference counterpart): build this batch's hit masks NOW, on a side stream -- they
        depend on the ids only -- and hand the result to ``forward(..., prepared=...)`` with the SAME
        ``item_idx`` / ``pos_idx`` tensors.  Takes ~35 us of small kernels off the step's critical path."""
        return prepare_masks(item_idx, pos_idx, batch_size=batch_size, embedding_dim=embedding_dim,
                             num_negatives=int(self.num_negatives), pos_csr=pos_csr)

    # ---- the reference's public helper methods, on caller-provided tensors (API parity; not the hot path)
    @torch.no_grad()
    def negative_masks(self, logits: torch.Tensor, *, item_idx: torch.Tensor,
                       pos_idx: torch.Tensor | None = None) -> torch.Tensor:
        """``~accidental_hits`` of losses.py:92-110: bool ``[B, N]``, True = valid negative."""
        b, n = logits.size(0), item_idx.numel()
        ii = _lib.dev_i64(item_idx, "item_idx")
        pi = None if pos_idx is None or pos_idx.shape[1] == 0 else _lib.dev_i64(pos_idx, "pos_idx")
        p = 0 if pi is None else pi.shape[1]
        lib = _lib.lib()
        ws = _lib.workspace(lib.mf_negative_masks_ws_bytes(b, n, p), ii.device)
        out = torch.empty(b, n, dtype=torch.uint8, device=ii.device)
        _lib.check(lib.mf_negative_masks(b, n, p, _lib.ptr(ii), _lib.ptr(pi), _lib.ptr(ws), ws.numel(), _lib.ptr(out),
                                         _lib.stream_ptr()))
        return out.bool()

    @torch.no_grad()
    def _mine(self, logits: torch.Tensor, negative_masks: torch.Tensor, semi_hard: bool) -> torch.Tensor:
        lg = _lib.dev_f32(logits, "logits")
        m = negative_masks.to(device=lg.device, dtype=torch.uint8).contiguous().clone()
        _lib.check(_lib.lib().mf_mine_logits(_lib.ptr(lg), lg.shape[0], lg.shape[1], int(self.num_negatives),
                                             int(semi_hard), _lib.ptr(m), _lib.stream_ptr()))
        return m.bool()

    def hard_mining(self, logits: torch.Tensor, negative_masks: torch.Tensor) -> torch.Tensor:
        """losses.py:112-132 (defined upstream, never called): keep the ``num_negatives`` highest logits."""
        return self._mine(logits, negative_masks, semi_hard=False)

    def semi_hard_mining(self, logits: torch.Tensor, negative_masks: torch.Tensor) -> torch.Tensor:
        """losses.py:134-162 on a materialised logits matrix (the losses themselves mine on the fly)."""
        return self._mine(logits, negative_masks, semi_hard=True)

    def _loss_of_kind(self, kind: int, user_embed, item_embed, target, *, item_idx, pos_idx=None, logq=None, prepared=None,
                      logq_table=None, pos_csr=None) -> torch.Tensor:
        k = int(self.num_negatives)
        if kind != 0:
            _check_num_negatives(k, item_embed.size(0))
        return _LossFunction.apply(user_embed, item_embed, target, item_idx, pos_idx, logq, 1 << kind, kind,
                                   k, float(self.sigma), float(self.margin), prepared, logq_table, None, pos_csr)

    def loss(self, user_embed, item_embed, target, *, item_idx, pos_idx=None, logq=None, prepared=None,
             logq_table=None, pos_csr=None) -> torch.Tensor:
        return self._loss_of_kind(self.kind, user_embed, item_embed, target, item_idx=item_idx, pos_idx=pos_idx, logq=logq,
                                  prepared=prepared, logq_table=logq_table, pos_csr=pos_csr)

    # ---- the reference's per-loss methods (losses.py:164-246): any instance can evaluate any of them, with ITS
    # num_negatives / sigma / margin, exactly like upstream where the subclasses only pick one
    def alignment_loss(self, user_embed: torch.Tensor, item_embed: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """losses.py:164-170 (positives only; no ids needed)."""
        ids = torch.zeros(item_embed.size(0), dtype=torch.int64, device=item_embed.device)
        return self._loss_of_kind(KINDS.index("AlignmentLoss"), user_embed, item_embed, target, item_idx=ids, pos_idx=None)

    def contrastive_loss(self, user_embed, item_embed, target, *, item_idx, pos_idx, **kw) -> torch.Tensor:
        """losses.py:172-193."""
        return self._loss_of_kind(KINDS.index("ContrastiveLoss"), user_embed, item_embed, target, item_idx=item_idx,
                                  pos_idx=pos_idx, **kw)

    def infonce_loss(self, user_embed, item_embed, target, *, item_idx, pos_idx, **kw) -> torch.Tensor:
        """losses.py:195-223."""
        return self._loss_of_kind(KINDS.index("InfomationNoiseContrastiveEstimationLoss"), user_embed, item_embed, target,
                                  item_idx=item_idx, pos_idx=pos_idx, **kw)

    def mine_loss(self, user_embed, item_embed, target, *, item_idx, pos_idx, **kw) -> torch.Tensor:
        """losses.py:225-246."""
        return self._loss_of_kind(KINDS.index("MutualInformationNeuralEstimationLoss"), user_embed, item_embed, target,
                                  item_idx=item_idx, pos_idx=pos_idx, **kw)


class AlignmentLoss(EmbeddingLoss):  # losses.py:249-259
    pass


class ContrastiveLoss(EmbeddingLoss):  # losses.py:262-274
    pass


class AlignmentContrastiveLoss(EmbeddingLoss):  # losses.py:277-291
    pass


class InfomationNoiseContrastiveEstimationLoss(EmbeddingLoss):  # losses.py:294-306 (spelling kept)
    pass


class MutualInformationNeuralEstimationLoss(EmbeddingLoss):  # losses.py:309-321
    pass


class PairwiseEmbeddingLoss(EmbeddingLoss, abc.ABC):  # losses.py:324-349
    @abc.abstractmethod
    def score_loss_fn(self, score: torch.Tensor) -> torch.Tensor:
        """phi of losses.py:348-349 on a caller's tensor (API parity; the kernels apply it per tile element)."""


class PairwiseLogisticLoss(PairwiseEmbeddingLoss):  # losses.py:352-354 (BPR when margin = 0)
    def score_loss_fn(self, score: torch.Tensor) -> torch.Tensor:
        return -torch.nn.functional.logsigmoid(-score)


class PairwiseHingeLoss(PairwiseEmbeddingLoss):  # losses.py:357-359 (default train_loss)
    def score_loss_fn(self, score: torch.Tensor) -> torch.Tensor:
        return score.relu()


@torch.no_grad()
def negative_mask(user_embed, item_embed, target, *, item_idx, pos_idx=None, num_negatives=0, sigma=1.0,
                  logq=None, pos_csr=None) -> torch.Tensor:
    """Boolean ``[B, N]`` mask of the negatives that enter the loss: ``negative_masks``
    followed by ``semi_hard_mining`` (losses.py:92-162), computed by the HIP path
    (``out_mask_bits`` of ``mf_loss_fwd``).  Diagnostic / test helper."""
    check_inputs(user_embed, item_embed, target)
    u, v, t, ii, pi, p, (lq, lq_rows), d, dp = _prepare(user_embed, item_embed, target, item_idx, pos_idx, logq)
    b, n = u.shape[0], v.shape[0]
    nw = (n + 31) // 32
    lib = _lib.lib()
    ws = _lib.workspace(lib.mf_loss_ws_bytes(b, n, dp, p, int(num_negatives)), u.device)
    out = torch.zeros(len(KINDS), dtype=torch.float32, device=u.device)
    bits = torch.zeros(b, nw, dtype=torch.int32, device=u.device)
    flags = _lib.LOSS_TARGET_I64 if t.dtype == torch.int64 else 0
    if pos_csr is None:
        _lib.check(lib.mf_loss_fwd(b, n, dp, p, int(num_negatives), float(sigma), 1.0, 1 << 1, _lib.ptr(u), _lib.ptr(v),
                                   _lib.ptr(t), _lib.ptr(ii), _lib.ptr(pi), _lib.ptr(lq), lq_rows, flags, _lib.ptr(ws), ws.numel(),
                                   _lib.ptr(out), _lib.ptr(bits), _lib.stream_ptr()))
    else:
        uid, off, items, nu = _csr_args(pos_csr, b)
        _lib.check(lib.mf_loss_fwd_csr(b, n, dp, int(num_negatives), float(sigma), 1.0, 1 << 1, _lib.ptr(u), _lib.ptr(v),
                                       _lib.ptr(t), _lib.ptr(ii), _lib.ptr(uid), _lib.ptr(off), _lib.ptr(items), nu, _lib.ptr(lq),
                                       lq_rows, flags, _lib.ptr(ws), ws.numel(), _lib.ptr(out), _lib.ptr(bits), _lib.stream_ptr()))
    shifts = torch.arange(32, device=u.device, dtype=torch.int32)
    return ((bits[:, :, None] >> shifts) & 1).bool().reshape(b, nw * 32)[:, :n]
