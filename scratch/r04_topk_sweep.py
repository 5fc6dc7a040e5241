"""Round 4: the retrieval-metrics pass (FactorizedTopK: ranks of the true items against the whole item corpus) against the corpus size and
the number of queries - MFMA time at the f32 peak beside it."""
import json, sys, time
import torch
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd import metrics
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer
dev = torch.device("cuda:0")
for n_items in [100_000, 1_000_000, 1_000_003, 1_048_576, 3_000_000]:
    for b in [8192, 8200, 1000]:
        cfg = TwoTowerConfig(n_users=1_000_000, n_items=n_items, embedding_dim=128, tower_dims=[256, 128], temperature=0.1,
                             l2_regularization=1e-6, learning_rate=0.001, optimizer="sgd", batch_size=b)
        tr = TwoTowerTrainer(cfg, dev, seed=1)
        corpus = tr.item_corpus_embeddings()
        m = metrics.FactorizedTopK(ks=(1, 5, 10, 100))
        u, i = tr.synthetic_batch(1, 0)
        for _ in range(3): tr.evaluate_topk(u, i, m, corpus)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 10
        for _ in range(n): tr.evaluate_topk(u, i, m, corpus)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        ideal = 2.0 * b * n_items * 128 / 157.3e12 * 1e3
        print(json.dumps({"items": n_items, "queries": b, "ms": round(ms, 3), "ideal_mfma_ms": round(ideal, 3), "frac": round(ideal / ms, 3)}), flush=True)
        del tr, corpus; torch.cuda.empty_cache()
