"""Round 4: batches that are balanced over the row ranges but full of repeated ids (tables much smaller than the batch): one-launch optimizer
against plan + optimizer step (TT_FUSE_SORT=0)."""
import json, os, sys, time
import torch
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer
dev = torch.device("cuda:0")
for rows in [(1000, 1000), (10_000, 10_000), (50_000, 50_000), (200_000, 100_000), (2_000_000, 1_000_000)]:
    cfg = TwoTowerConfig(n_users=rows[0], n_items=rows[1], embedding_dim=128, tower_dims=[256, 128], temperature=0.1,
                         l2_regularization=1e-6, learning_rate=0.001, optimizer="sgd", batch_size=8192)
    tr = TwoTowerTrainer(cfg, dev, seed=1)
    batches = [tr.synthetic_batch(1, s) for s in range(8)]
    for s in range(30): tr.step(*batches[s % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 150
    for s in range(n): tr.step(*batches[s % 8])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"rows": rows, "fuse_sort": tr.fuse_sort, "ms_per_step": round(ms, 4), "range_load": tr.range_load,
                      "one_launch": tr.one_launch_optimizer(8192)}), flush=True)
    del tr; torch.cuda.empty_cache()
