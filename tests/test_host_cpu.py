"""Host-side logic that needs no GPU: the reference's YAML schema, the parquet reader, the task object's
argument validation (error behaviour mirrors tfrs.tasks.Retrieval)."""
import numpy as np
import pandas as pd
import pytest
import yaml

from two_tower_amazon_recommender_amd import config as cfgmod, data as datamod
from two_tower_amazon_recommender_amd.tasks import Retrieval
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer

def reference_model_block() -> dict:
    """The hyper-parameter contract the reference states in the `model:` block of configs/data_config.yaml:54-71,
    as values (key names and numbers are the schema this repo must read; the file itself is not copied)."""
    towers = [512, 256, 128]
    return {"model": {
        "embedding_dim": 128, "user_tower_dims": list(towers), "item_tower_dims": list(towers),
        "dropout_rate": 0.1, "l2_regularization": 1e-6,
        "training": {"batch_size": 1024, "learning_rate": 0.001, "epochs": 50, "patience": 5, "validation_freq": 1},
        "retrieval": {"candidate_sampling": "in_batch", "temperature": 0.1, "top_k_eval": [1, 5, 10, 20, 50, 100]},
    }}


def test_reference_yaml_schema_is_read_verbatim():
    doc = yaml.safe_load(yaml.safe_dump(reference_model_block()))      # through YAML, as load_yaml would deliver it
    cfg, loop = cfgmod.model_config_from_dict(doc, 1000, 2000, dropout_override=0.0)
    assert (cfg.embedding_dim, cfg.tower_dims, cfg.batch_size) == (128, [512, 256, 128], 1024)
    assert cfg.temperature == 0.1 and cfg.learning_rate == 0.001 and cfg.l2_regularization == 1e-6
    assert loop == dict(epochs=50, patience=5, validation_freq=1, top_k_eval=[1, 5, 10, 20, 50, 100])
    cfg.validate()
    cfg2, _ = cfgmod.model_config_from_dict(doc, 1000, 2000)
    assert cfg2.dropout_rate == 0.1
    cfg2.validate()
    doc["model"]["retrieval"]["candidate_sampling"] = "uniform"
    with pytest.raises(NotImplementedError, match="in_batch"):
        cfgmod.model_config_from_dict(doc, 10, 10)


def test_parquet_reader_accepts_both_reference_encodings(tmp_path):
    df = pd.DataFrame({"user_id": list("abca"), "parent_asin": list("xyzx"), "rating": [5.0, 4.0, 3.0, 1.0],
                       "user_idx": np.array([0, 1, 2, 0]), "item_idx": np.array([0, 1, 2, 0])})
    p = tmp_path / "combined_interactions.parquet"
    df.to_parquet(p, compression="snappy", index=False)           # prepare_training_data.py:218
    u, i = datamod.read_interactions(p)
    assert u.dtype == np.int64 and np.array_equal(u, [0, 1, 2, 0]) and np.array_equal(i, [0, 1, 2, 0])
    df2 = df.rename(columns={"user_idx": "user_id_encoded", "item_idx": "item_id_encoded"})
    p2 = tmp_path / "enc.parquet"
    df2.to_parquet(p2, index=False)
    assert np.array_equal(datamod.read_interactions(p2)[0], [0, 1, 2, 0])
    df.drop(columns=["user_idx"]).to_parquet(tmp_path / "bad.parquet", index=False)
    with pytest.raises(KeyError):
        datamod.read_interactions(tmp_path / "bad.parquet")


def test_category_column_reader_and_batch_triples(tmp_path):
    """The hashed-category input: `category` strings (prepare_training_data.py:47), nulls -> "Unknown"
    (preprocessor.py:480), integer `category_encoded` by its decimal text; the iterator yields triples."""
    df = pd.DataFrame({"user_idx": np.arange(6), "item_idx": np.arange(6)[::-1].copy(),
                       "category": ["Books", "Electronics", None, "Books", "All_Beauty", "Electronics"]})
    p = tmp_path / "c.parquet"
    df.to_parquet(p, index=False)
    codes, values = datamod.read_category_values(p)
    assert [values[c] for c in codes] == ["Books", "Electronics", "Unknown", "Books", "All_Beauty", "Electronics"]
    df2 = df.drop(columns=["category"]).assign(category_encoded=np.array([3, 1, 3, 0, 1, 1]))
    p2 = tmp_path / "e.parquet"
    df2.to_parquet(p2, index=False)
    codes2, values2 = datamod.read_category_values(p2)
    assert [values2[c] for c in codes2] == ["3", "1", "3", "0", "1", "1"]
    df.drop(columns=["category"]).to_parquet(tmp_path / "n.parquet", index=False)
    assert datamod.read_category_values(tmp_path / "n.parquet") is None
    it = datamod.BatchIterator(df.user_idx.to_numpy(), df.item_idx.to_numpy(), 2, "cpu", shuffle=True,
                               category_bucket=np.array([7, 8, 9, 7, 5, 8]))
    seen = [(int(u), int(c)) for b in it for u, c in zip(b[0], b[2])]
    assert sorted(seen) == [(0, 7), (1, 8), (2, 9), (3, 7), (4, 5), (5, 8)]        # the triple stays aligned under shuffling
    with pytest.raises(ValueError):
        datamod.BatchIterator(np.arange(4), np.arange(4), 2, "cpu", category_bucket=np.arange(3))


def test_retrieval_task_argument_errors_mirror_tfrs():
    with pytest.raises(ValueError):
        Retrieval(num_hard_negatives=0)
    with pytest.raises(NotImplementedError):
        Retrieval(metrics=object())
    with pytest.raises(ValueError):
        Retrieval(temperature=0.0)
    task = Retrieval(temperature=0.1, remove_accidental_hits=True)
    import torch
    q = torch.zeros(4, 32)
    with pytest.raises(ValueError, match="candidate ids must be supplied"):
        task(q, q)


def test_trainer_refuses_cpu_device():
    cfg = TwoTowerConfig(n_users=10, n_items=10, embedding_dim=32, tower_dims=[32], batch_size=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TwoTowerTrainer(cfg, device="cpu")
    with pytest.raises(ValueError):
        TwoTowerConfig(n_users=10, n_items=10, embedding_dim=32, tower_dims=[48], batch_size=8).validate()
