import json, sys, time, torch
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer
dev = torch.device("cuda:0")
for item_dims in (None, [512, 128], [256, 256, 128]):
    cfg = TwoTowerConfig(n_users=2_000_000, n_items=1_000_000, embedding_dim=128, tower_dims=[256, 128], item_tower_dims=item_dims, temperature=0.1,
                         l2_regularization=1e-6, learning_rate=0.001, optimizer="sgd", batch_size=8192)
    tr = TwoTowerTrainer(cfg, dev, seed=1)
    bs = [tr.synthetic_batch(1, s) for s in range(8)]
    for s in range(30): tr.step(*bs[s % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 150
    for s in range(n): tr.step(*bs[s % 8])
    torch.cuda.synchronize()
    print(json.dumps({"item_tower_dims": item_dims, "symmetric": cfg.symmetric, "ms_per_step": round((time.perf_counter() - t0) / n * 1e3, 4)}), flush=True)
    del tr; torch.cuda.empty_cache()
