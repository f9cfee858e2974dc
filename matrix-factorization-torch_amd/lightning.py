"""``MatrixFactorizationLitModule`` on the HIP hot path.

Mirrors the surface of ``xfmr_rec/lightning.py`` that the training loop touches:
``MatrixFactorizationLitConfig`` (:32-43, same defaults), ``forward`` (:60-74),
``recommend`` (:76-95), ``compute_losses`` (:97-147, same ``"{step}/{ClassName}"`` keys,
all seven losses per step), ``training_step`` (:189-192), ``configure_optimizers``
(:238-239), ``configure_model`` / ``get_loss_fns`` (:252-287) and the
``"... must be initialised first"`` guards (:61-63, :85-87, :100-102).

Differences, all dictated by the north-star: towers are embedding tables indexed by
``user_rn`` / ``movie_rn`` (``forward(idx, tower=...)`` instead of ``forward(text)``),
the optimiser is a sparse row optimiser, retrieval is exact brute force.  The class
derives from ``lightning.LightningModule`` when Lightning is installed (it is not in the
build image) and from ``torch.nn.Module`` otherwise, so ``Trainer.fit`` can drive it
unchanged where Lightning exists.  The retrieval metrics (``validation_step`` / ``test_step`` /
``predict_step``, on the device), ``save`` / ``load`` and the two id-based serving entry points ARE
implemented; callbacks, loggers and the CLI are the reference's control
plane and stay out of scope.
"""
from __future__ import annotations

import torch

from . import losses as mf_losses
from . import fused, models, optim
from .params import INDEX_PATH, PROCESSORS_JSON, TOP_K, TOWERS_PATH
from .retrieval import RetrievalMetrics, ItemProcessor

try:  # pragma: no cover - Lightning is absent from the build image
    from lightning import LightningModule as _Base
except ImportError:  # noqa: SIM105
    class _Base(torch.nn.Module):
        """Minimal stand-in: what this module uses of LightningModule."""

        def log_dict(self, *_, **__) -> None:
            return None

        @property
        def device(self) -> torch.device:
            return next(self.parameters()).device


class MatrixFactorizationLitConfig(models.ModelConfig):
    hidden_size: int = 32          # xfmr_rec/lightning.py:33
    train_loss: str = "PairwiseHingeLoss"
    num_negatives: int = 4
    sigma: float = 1.0
    margin: float = 1.0
    learning_rate: float = 0.0001
    top_k: int = TOP_K
    optimizer: str = "adam"        # "adam" = row-wise AdamW (the reference's optimiser class), "sgd"
    fused_losses: bool = True      # all seven losses from one sweep instead of seven sweeps
    use_logq: bool = False         # logQ correction (absent upstream)


class MatrixFactorizationLitModule(_Base):
    def __init__(self, config: MatrixFactorizationLitConfig | dict) -> None:
        super().__init__()
        self.config = MatrixFactorizationLitConfig.model_validate(config)
        self.towers: torch.nn.ModuleDict | None = None
        self.loss_fns: torch.nn.ModuleList | None = None
        self.item_processor: ItemProcessor | None = None
        self.history: dict[int, list[int]] = {}      # user_rn -> item ids already consumed (recommend excludes them)
        self.logq: torch.Tensor | None = None         # [num_items] log sampling probability, when use_logq
        self.metrics: dict[str, RetrievalMetrics] | None = None
        self._fused: fused.FusedSmallStep | None = None

    # ------------------------------------------------------------------ towers ---
    def forward(self, idx: torch.Tensor, *, tower: str = "user") -> torch.Tensor:
        if self.towers is None:
            msg = "`model` must be initialised first"
            raise ValueError(msg)
        return self.towers[tower](idx)

    @torch.inference_mode()
    def recommend(self, user_idx: int, *, top_k: int = TOP_K, exclude_item_ids: list[int] | None = None):
        if self.towers is None or self.item_processor is None or self.item_processor.index is None:
            msg = "`user_processor` and `item_processor` must be initialised first"
            raise ValueError(msg)
        exclude_item_ids = (exclude_item_ids or []) + list(self.history.get(int(user_idx), []))
        device = self.towers["user"].weight.device
        embed = self(torch.tensor([int(user_idx)], device=device)).cpu().numpy()
        return self.item_processor.search(embed, exclude_item_ids=exclude_item_ids, top_k=top_k)

    @torch.inference_mode()
    def recommend_with_item_id(self, item_id: int, *, top_k: int = TOP_K, exclude_item_ids: list[int] | None = None):
        """Items closest to item ``item_id`` -- the query is the item's own embedding and the item itself is
        excluded (``Service.recommend_with_item_id`` -> ``recommend_with_item``, xfmr_rec/bentoml/service.py:220-247)."""
        if self.item_processor is None or self.item_processor.index is None:
            msg = "`user_processor` and `item_processor` must be initialised first"
            raise ValueError(msg)
        row = self.item_processor.row_of(item_id)
        embed = self.item_processor.index.embeddings[row: row + 1, : self.config.hidden_size].cpu().numpy()
        return self.item_processor.search(embed, exclude_item_ids=[*(exclude_item_ids or []), int(item_id)], top_k=top_k)

    def recommend_with_user_id(self, user_idx: int, *, top_k: int = TOP_K, exclude_item_ids: list[int] | None = None):
        """``Service.recommend_with_user_id`` (service.py:286-311): the user's history is excluded."""
        return self.recommend(user_idx, top_k=top_k, exclude_item_ids=exclude_item_ids)

    # ------------------------------------------------------------- save / load (f-3) ---
    def save(self, path) -> None:
        """The counterpart of ``save`` (xfmr_rec/lightning.py:312-328: ``transformer/`` + ``processors.json`` +
        ``lance_db/``): the two tables (``towers.safetensors``), ``processors.json`` (config, item ids,
        histories) and the built item index (``item_index.safetensors``: unit-norm item matrix + id map)."""
        import json
        import pathlib

        from safetensors.torch import save_file

        if self.towers is None:
            msg = "`model` must be initialised first"
            raise ValueError(msg)
        path = pathlib.Path(path)
        path.mkdir(parents=True, exist_ok=True)
        save_file({f"{k}.weight": t.weight.detach().cpu().contiguous() for k, t in self.towers.items()}, str(path / TOWERS_PATH))
        proc = {"config": self.config.model_dump(), "history": {str(k): list(map(int, v)) for k, v in self.history.items()}}
        (path / PROCESSORS_JSON).write_text(json.dumps(proc, indent=2))
        if self.item_processor is not None and self.item_processor.index is not None:
            idx = self.item_processor.index
            save_file({"embeddings": idx.embeddings[:, : idx.dim].cpu().contiguous(),
                       "item_ids": self.item_processor.item_ids.contiguous()}, str(path / INDEX_PATH))

    @classmethod
    def load(cls, path, device="cuda") -> "MatrixFactorizationLitModule":
        """What serving loads (``bentoml/service.py`` builds its processors from the same three pieces)."""
        import json
        import pathlib

        from safetensors.torch import load_file

        path = pathlib.Path(path)
        proc = json.loads((path / PROCESSORS_JSON).read_text())
        module = cls(proc["config"])
        module.configure_model(device=device)
        weights = load_file(str(path / TOWERS_PATH))
        with torch.no_grad():
            for k, t in module.towers.items():
                t.weight.copy_(weights[f"{k}.weight"].to(device))
        module.history = {int(k): v for k, v in proc.get("history", {}).items()}
        if (path / INDEX_PATH).exists():
            idx = load_file(str(path / INDEX_PATH))
            module.item_processor = ItemProcessor(idx["item_ids"])
            module.item_processor.set_index(idx["embeddings"].to(device))
        return module

    # ------------------------------------------------------------------ losses ---
    def compute_losses(self, batch, step_name: str = "train") -> dict[str, torch.Tensor]:
        if self.loss_fns is None:
            msg = "`loss_fns` must be initialised first"
            raise ValueError(msg)
        target = batch["target"]
        pos_idx = batch["user"].get("pos_idx")            # the reference's padded positives, or ...
        pos_csr = batch["user"].get("pos_csr")            # ... the producer's CSR lists (data.DeviceInteractionSampler)
        user_embed = self(batch["user"]["idx"], tower="user")
        # positives then sampled negatives, as xfmr_rec/lightning.py:133-134
        item_idx = torch.cat([batch["item"]["idx"], batch["neg_item"]["idx"]])
        item_embed = self(item_idx, tower="item")
        # the logQ table is looked up by item_idx inside the kernel (no gather launch)
        logq_table = self.logq if (self.config.use_logq and self.logq is not None) else None
        cfg = self.config
        if cfg.fused_losses:
            vals = mf_losses.fused_losses(user_embed, item_embed, target, item_idx=item_idx, pos_idx=pos_idx,
                                          num_negatives=cfg.num_negatives, sigma=cfg.sigma, margin=cfg.margin,
                                          logq_table=logq_table, pos_csr=pos_csr,
                                          train_loss=cfg.train_loss if step_name == "train" else None)
            return {f"{step_name}/{name}": v for name, v in vals.items()}
        return {
            f"{step_name}/{fn.__class__.__name__}": fn(user_embed=user_embed, item_embed=item_embed, target=target,
                                                       item_idx=item_idx, pos_idx=pos_idx, logq_table=logq_table, pos_csr=pos_csr)
            for fn in self.loss_fns
        }

    # ------------------------------------------------------------------ metrics ---
    @torch.no_grad()
    def update_metrics(self, batch, step_name: str = "val") -> dict[str, torch.Tensor]:
        """Batched counterpart of ``update_metrics`` (xfmr_rec/lightning.py:149-187, one example per
        call upstream): top-k for all users of the batch with their histories excluded
        (``predict_step`` -> ``recommend``, :76-95), then the six retrieval metrics on the device.
        ``batch``: ``user.idx`` [Q]; ``history`` = (offsets [Q+1], item rows) to exclude; ``target`` =
        (offsets [Q+1], item rows, ratings) -- item rows are the index's ``movie_rn``."""
        if self.metrics is None:
            msg = "`metrics` must be initialised first"
            raise ValueError(msg)
        if self.item_processor is None or self.item_processor.index is None:
            msg = "`user_processor` and `item_processor` must be initialised first"
            raise ValueError(msg)
        queries = self(batch["user"]["idx"], tower="user")
        _, rows = self.item_processor.index.search(queries, self.config.top_k, exclude_csr=batch.get("history"))
        off, ids, rating = batch["target"]
        metric = self.metrics[step_name]
        metric.update(rows, off, ids, rating)
        return metric.compute()

    def validation_step(self, batch, _: int = 0) -> None:
        self.log_dict(self.update_metrics(batch, step_name="val"))

    def test_step(self, batch, _: int = 0) -> None:
        self.log_dict(self.update_metrics(batch, step_name="test"))

    @torch.no_grad()
    def predict_step(self, batch, _: int = 0) -> tuple[torch.Tensor, torch.Tensor]:
        """``(scores, item rows)`` [Q, top_k] of the batch's users, histories excluded."""
        queries = self(batch["user"]["idx"], tower="user")
        return self.item_processor.index.search(queries, self.config.top_k, exclude_csr=batch.get("history"))

    def training_step(self, batch, _: int = 0) -> torch.Tensor:
        losses = self.compute_losses(batch, step_name="train")
        self.log_dict(losses)
        return losses[f"train/{self.config.train_loss}"]

    def fused_training_step(self, batch, optimizer: torch.optim.Optimizer) -> torch.Tensor:
        """``training_step`` + ``loss.backward()`` + ``optimizer.step()`` (xfmr_rec/lightning.py:189-192 and Lightning's
        automatic optimisation around it) in ONE launch when the configuration is the reference's kind -- B <= 128 pairs,
        plain embedding towers, a mined loss, ``optim.SparseSGD`` / ``optim.RowAdam`` -- through ``fused.FusedSmallStep``
        (``mf_step_small``: bit-identical to the three calls); any other configuration runs those three calls.  Returns
        the trained loss, detached (its gradient has been applied); logs all seven losses like ``training_step``."""
        if self.loss_fns is None or self.towers is None:
            msg = "`loss_fns` must be initialised first"
            raise ValueError(msg)
        cfg = self.config
        if self._fused is None or self._fused.opt is not optimizer:
            fn = next(f for f in self.loss_fns if f.__class__.__name__ == cfg.train_loss)
            try:
                self._fused = fused.FusedSmallStep(self.towers, optimizer, fn, all_losses=True,
                                                   logq_table=self.logq if (cfg.use_logq and self.logq is not None) else None)
            except mf_losses._lib.MfHipError:        # hash towers, a foreign optimiser: the ordinary three calls
                self._fused = None
        if self._fused is not None:                  # the table of THIS step (it may be set / refreshed after the first one)
            self._fused.set_logq_table(self.logq if (cfg.use_logq and self.logq is not None) else None)
        if self._fused is None:
            loss = self.training_step(batch)
            loss.backward()
            optimizer.step()
            optimizer.zero_grad(set_to_none=True)
            return loss.detach()
        flat = {"user": batch["user"]["idx"], "item": torch.cat([batch["item"]["idx"], batch["neg_item"]["idx"]]), "target": batch["target"]}
        if batch["user"].get("pos_csr") is not None:
            flat["pos_csr"] = (flat["user"],) + tuple(batch["user"]["pos_csr"][1:])
        elif batch["user"].get("pos_idx") is not None:
            flat["pos"] = batch["user"]["pos_idx"]
        before = self._fused.fused_steps
        loss = self._fused(flat)
        if self._fused.fused_steps > before:       # (a fallback step evaluates the trained loss only)
            self.log_dict({f"train/{name}": self._fused.losses[i] for i, name in enumerate(mf_losses.KINDS)})
        return loss

    # ------------------------------------------------------------------- setup ---
    def configure_optimizers(self) -> torch.optim.Optimizer:
        params = list(self.towers.parameters())
        if self.config.optimizer == "sgd":
            return optim.SparseSGD(params, lr=self.config.learning_rate)
        return optim.RowAdam(params, lr=self.config.learning_rate)

    def configure_model(self, device=None) -> None:
        if self.towers is None:
            self.towers = self.get_model(device)
        if self.loss_fns is None:
            self.loss_fns = self.get_loss_fns()
        if self.item_processor is None:
            self.item_processor = ItemProcessor()
        if self.metrics is None:
            self.metrics = self.get_metrics()

    def get_model(self, device=None) -> torch.nn.ModuleDict:
        return models.init_towers(self.config, device=device)

    def get_loss_fns(self) -> torch.nn.ModuleList:
        loss_classes = [
            mf_losses.AlignmentLoss,
            mf_losses.ContrastiveLoss,
            mf_losses.AlignmentContrastiveLoss,
            mf_losses.InfomationNoiseContrastiveEstimationLoss,
            mf_losses.MutualInformationNeuralEstimationLoss,
            mf_losses.PairwiseHingeLoss,
            mf_losses.PairwiseLogisticLoss,
        ]
        cfg = self.config
        return torch.nn.ModuleList(
            [cls(num_negatives=cfg.num_negatives, sigma=cfg.sigma, margin=cfg.margin) for cls in loss_classes]
        )

    def get_metrics(self) -> dict[str, RetrievalMetrics]:
        """NDCG / Recall / Precision / MAP / HitRate / MRR @top_k for "val" and "test" (lightning.py:289-306)."""
        return {step: RetrievalMetrics(top_k=self.config.top_k, prefix=f"{step}/") for step in ("val", "test")}

    def on_validation_start(self) -> None:
        self.item_processor.get_index(self)
        self.metrics["val"].reset()

    def on_test_start(self) -> None:
        self.item_processor.get_index(self)
        self.metrics["test"].reset()
