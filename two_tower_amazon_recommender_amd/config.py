"""Reads the reference's hyper-parameter schema: the ``model:`` block of
``/root/reference/configs/data_config.yaml:54-71`` (no code in the reference reads it)."""
from __future__ import annotations

import yaml

from .trainer import TwoTowerConfig


def load_yaml(path) -> dict:
    with open(path, "r") as f:           # yaml.safe_load, as download_data.py:32-39 does
        return yaml.safe_load(f)


def model_config_from_dict(doc: dict, n_users: int, n_items: int, optimizer: str = "adagrad",
                           dropout_override: float | None = None) -> tuple[TwoTowerConfig, dict]:
    """Returns (TwoTowerConfig, training-loop settings {epochs, patience, validation_freq, top_k_eval})."""
    m = doc.get("model")
    if not isinstance(m, dict):
        raise KeyError("config has no 'model:' block (configs/data_config.yaml:54)")
    user_dims, item_dims = list(m["user_tower_dims"]), list(m["item_tower_dims"])
    tr, rt = m.get("training", {}), m.get("retrieval", {})
    sampling = rt.get("candidate_sampling", "in_batch")
    if sampling != "in_batch":
        raise NotImplementedError(f"candidate_sampling {sampling!r}: only 'in_batch' is implemented")
    dropout = float(m.get("dropout_rate", 0.0)) if dropout_override is None else dropout_override
    cfg = TwoTowerConfig(
        n_users=n_users, n_items=n_items, embedding_dim=int(m["embedding_dim"]), tower_dims=user_dims,
        item_tower_dims=None if item_dims == user_dims else item_dims,
        temperature=float(rt.get("temperature", 1.0)), l2_regularization=float(m.get("l2_regularization", 0.0)),
        learning_rate=float(tr.get("learning_rate", 0.001)), optimizer=optimizer,
        batch_size=int(tr.get("batch_size", 1024)), dropout_rate=dropout)
    loop = dict(epochs=int(tr.get("epochs", 1)), patience=int(tr.get("patience", 5)),
                validation_freq=int(tr.get("validation_freq", 1)), top_k_eval=list(rt.get("top_k_eval", [])))
    return cfg, loop
