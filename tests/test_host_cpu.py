"""Host-side logic that needs no GPU: the reference's YAML schema, the parquet reader, the task object's
argument validation (error behaviour mirrors tfrs.tasks.Retrieval)."""
import os
import pathlib

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
    with pytest.raises(TypeError):
        Retrieval(metrics=object())                # only metrics.FactorizedTopK(candidates=...) is a metric the task can run
    with pytest.raises(TypeError):
        Retrieval(batch_metrics=[object()])        # only metrics.TopKCategoricalAccuracy (rank < k from the fused rank pass);
                                                   # anything else would need the materialised in-batch score matrix
    from two_tower_amazon_recommender_amd.metrics import TopKCategoricalAccuracy
    from two_tower_amazon_recommender_amd.tasks import CategoricalCrossentropy
    assert len(Retrieval(batch_metrics=[TopKCategoricalAccuracy(1), TopKCategoricalAccuracy(5)]).batch_metrics) == 2
    Retrieval(loss=CategoricalCrossentropy(from_logits=True, reduction="SUM"))       # the TFRS default, spelled out
    with pytest.raises(NotImplementedError, match="score matrix"):
        Retrieval(loss=CategoricalCrossentropy(from_logits=False))
    with pytest.raises(NotImplementedError):
        Retrieval(loss=object())
    with pytest.raises(TypeError):
        Retrieval(loss_metrics=[object()])
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


def test_early_stopping_follows_the_patience_of_the_reference_schema():
    """configs/data_config.yaml:65 `patience: 5`: stop after 5 evaluations in a row without improvement."""
    from two_tower_amazon_recommender_amd.train import EarlyStopping
    es = EarlyStopping(patience=3)
    assert [es.update(v) for v in (5.0, 4.0, 4.5, 4.2, 3.9)] == [False] * 5 and es.best == 3.9 and es.bad == 0
    assert [es.update(v) for v in (3.9, 3.95, 3.8999995)] == [False, False, True]     # within min_delta is no improvement
    es2 = EarlyStopping(patience=3)
    es2.load_state_dict(es.state_dict())
    assert (es2.best, es2.bad) == (es.best, es.bad) and es2.update(9.9)
    one = EarlyStopping(patience=1)
    assert one.update(1.0) is False and one.update(1.0) is True


def test_bench_self_launches_ranks_and_relays_one_json_line(tmp_path):
    """`python bench.py --gpus N` from a bare shell starts N ranks through torch.distributed.run as a child process,
    prints rank 0's JSON line on stdout and exits with the child's code — exercised here with a stand-in rank script
    (gloo, no GPU)."""
    import json
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parents[1]
    script = tmp_path / "fake_rank.py"
    script.write_text(
        "import json, os, sys\n"
        "import torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "print('noise from rank', r)\n"
        "dist.barrier()\n"
        "if r == 0: print(json.dumps({'value': 1.5, 'n_gpus': w, 'argv': sys.argv[1:]}))\n"
        "dist.destroy_process_group()\n"
        "sys.exit(int(os.environ.get('FAKE_RC', '0')) if r == w - 1 else 0)\n")
    driver = ("import sys; sys.path.insert(0, %r); import bench; sys.argv = ['bench.py', '--gpus', '2', '--steps', '3'];\n"
              "bench.__file__ = %r; args = bench.parse(); raise SystemExit(bench.self_launch(args))") % (str(root), str(script))
    for rc in (0, 3):
        env = dict(os.environ, FAKE_RC=str(rc))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        res = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, env=env, timeout=240)
        lines = [l for l in res.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, (res.stdout, res.stderr[-2000:])
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["argv"] == ["--gpus", "2", "--steps", "3"]
        assert (res.returncode == 0) == (rc == 0), (rc, res.returncode, res.stderr[-2000:])


def test_custom_ops_are_registered_with_schemas_and_fake_implementations():
    """north_star: "exposed to Python through PyTorch-ROCm custom ops".  Every hot-path op is a torch.library custom op
    (torch.ops.twotower.*) with a fake (meta) implementation, so shapes/dtypes propagate without a GPU; the real
    implementations are CUDA-only (no CPU fallback: a CPU tensor is refused)."""
    import torch
    from two_tower_amazon_recommender_amd import torch_ops
    for name in torch_ops.OPS:
        assert hasattr(torch.ops.twotower, name), name
    q, c = torch.empty(8, 32, device="meta"), torch.empty(16, 32, device="meta")
    loss, per, dq, dc = torch.ops.twotower.retrieval_loss(q, c, None, None, None, 10.0, 0, 0)
    assert loss.shape == () and per.shape == (8,) and dq.shape == (8, 32) and dc.shape == (16, 32)
    assert torch.ops.twotower.retrieval_rank(q, c, torch.empty(8, dtype=torch.int64, device="meta"), None, 1.0).dtype == torch.int32
    assert torch.ops.twotower.embedding_gather(torch.empty(100, 32, device="meta"), torch.empty(7, dtype=torch.int64, device="meta")).shape == (7, 32)
    y = torch.ops.twotower.dense_fwd(q, torch.empty(32, 64, device="meta"), None, True)
    assert y.shape == (8, 64)
    schema = str(torch.ops.twotower.sparse_update_.default._schema)
    assert "Tensor(a0!) table" in schema and "Tensor(a1!)? accum" in schema          # declared as mutating its table / accumulator
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.twotower.retrieval_loss(torch.zeros(4, 32), torch.zeros(4, 32), None, None, None, 1.0, 0, 0)   # CPU: no kernel
