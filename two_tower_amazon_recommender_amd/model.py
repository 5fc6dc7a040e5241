"""``TwoTowerModel`` — a ``tfrs.Model``-shaped facade over the HIP train step (the reference intended a
``tfrs.Model`` subclass in ``src/models``; it is a docstring stub, ``/root/reference/src/models/__init__.py:1``).

    model = TwoTowerModel(cfg)                       # cfg from config.model_config_from_dict(...)
    loss  = model.train_step({"user_idx": u, "item_idx": i})     # tfrs.Model.train_step(features)
    val   = model.test_step({"user_idx": u, "item_idx": i})
"""
from __future__ import annotations

import torch

from .trainer import TwoTowerConfig, TwoTowerTrainer

USER_KEYS = ("user_idx", "user_id_encoded")      # prepare_training_data.py:209 / preprocessor.py:481
ITEM_KEYS = ("item_idx", "item_id_encoded")      # :210 / :482
# hashed category feature (cfg.n_category_buckets > 0): int64 bucket ids under "category_bucket", or the raw
# category strings (a list of str) under "category" / "main_category", hashed on the GPU
CATEGORY_KEYS = ("category", "main_category")


def _pick(features: dict, keys):
    for k in keys:
        if k in features:
            return features[k]
    raise KeyError(f"features need one of {keys}")


class TwoTowerModel:
    def __init__(self, cfg: TwoTowerConfig, device="cuda:0", seed: int | None = 42):
        self.trainer = TwoTowerTrainer(cfg, device, seed=seed)
        self.cfg = cfg

    def compute_loss(self, features: dict, training: bool = False) -> torch.Tensor:
        u, i = _pick(features, USER_KEYS), _pick(features, ITEM_KEYS)
        kw = {k: features[k] for k in ("sample_weight", "candidate_sampling_probability", "candidate_ids") if k in features}
        if self.cfg.n_category_buckets:
            if "category_bucket" in features:
                kw["category_ids"] = features["category_bucket"]
            else:
                from . import ops
                rows = ops.strings_to_padded_bytes(list(_pick(features, CATEGORY_KEYS))).to(self.trainer.dev)
                kw["category_ids"] = ops.hash_buckets(rows, self.cfg.n_category_buckets)
        if training:
            return self.trainer.step(u, i, **kw)
        return self.trainer.evaluate(u, i, **kw)

    def train_step(self, features: dict, report_regularization: bool = True) -> dict:
        """tfrs.Model.train_step's dict: ``total_loss = loss + regularization_loss`` (the L2 terms of the Dense kernels,
        ``l2_regularization * sum(W^2)``, evaluated on the weights the step STARTED from, as Keras's ``model.losses`` are).
        The gradient of that term is fused into the dense update either way; ``report_regularization=False`` skips the
        reporting-only reduction (one small launch) and returns ``regularization_loss: None, total_loss: loss``."""
        reg = self.trainer.l2_penalty() if (report_regularization and self.cfg.l2_regularization > 0) else None
        loss = self.compute_loss(features, training=True)
        if reg is None:
            zero = report_regularization and self.cfg.l2_regularization == 0
            return {"loss": loss, "regularization_loss": torch.zeros_like(loss) if zero else None, "total_loss": loss}
        reg = reg.to(loss.dtype).reshape(loss.shape)
        return {"loss": loss, "regularization_loss": reg, "total_loss": loss + reg}

    def test_step(self, features: dict) -> dict:
        return {"loss": self.compute_loss(features, training=False)}
