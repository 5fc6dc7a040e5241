"""Round 4: step time over embedding dims / tower shapes / optimizers the bench lines never varied (batch 8192, 2M x 1M rows)."""
import json, sys, time
import torch
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer
dev = torch.device("cuda:0")
cases = [(128, [256, 128], "sgd", 0.0, 0), (128, [256, 128], "adagrad", 0.0, 0), (128, [256, 128], "sgd", 0.1, 0), (128, [256, 128], "adagrad", 0.0, 30),
         (64, [128, 64], "sgd", 0.0, 0), (32, [64, 32], "sgd", 0.0, 0), (256, [512, 256], "sgd", 0.0, 0), (128, [128], "sgd", 0.0, 0),
         (128, [512, 256, 128], "sgd", 0.0, 0), (128, [256, 256, 128], "sgd", 0.0, 0), (64, [256, 128], "sgd", 0.0, 0), (128, [256, 64], "sgd", 0.0, 0),
         (96, [192, 96], "sgd", 0.0, 0), (128, [320, 128], "sgd", 0.0, 0), (128, [256, 128, 128, 128], "sgd", 0.0, 0)]
b = 8192
for dim, towers, opt, drop, buckets in cases:
    try:
        cfg = TwoTowerConfig(n_users=2_000_000, n_items=1_000_000, embedding_dim=dim, tower_dims=towers, temperature=0.1,
                             l2_regularization=1e-6, learning_rate=0.001, optimizer=opt, batch_size=b, dropout_rate=drop, n_category_buckets=buckets)
        tr = TwoTowerTrainer(cfg, dev, seed=1)
        batches = [tr.synthetic_batch(1, s) for s in range(8)]
        cats = [tr.synthetic_categories(1, s) for s in range(8)] if buckets else None
        def step(s):
            kw = {"category_ids": cats[s % 8]} if buckets else {}
            tr.step(*batches[s % 8], **kw)
        for s in range(30): step(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 150
        for s in range(n): step(s)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        flops_t = 3 * 2 * 2 * b * sum(i * o for i, o in zip([dim] + towers[:-1], towers))
        flops_s = 6 * b * b * towers[-1]
        print(json.dumps({"dim": dim, "towers": towers, "opt": opt, "dropout": drop, "buckets": buckets, "ms_per_step": round(ms, 4),
                          "tower_GF": round(flops_t / 1e9, 2), "scorer_GF": round(flops_s / 1e9, 1), "TF_total": round((flops_t + flops_s) / ms / 1e9, 1),
                          "composite": tr._cstep is not None}), flush=True)
        del tr
    except Exception as e:
        print(json.dumps({"dim": dim, "towers": towers, "error": str(e)[:200]}), flush=True)
    torch.cuda.empty_cache()
