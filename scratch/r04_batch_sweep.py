"""Round 4: step time against the batch size around the headline's 8192 - which fused forms a ragged batch loses (fused tower forward,
dW tiles fed from global memory, fused two-tile dx workgroups, row-range id lists) and what that costs."""
import json, sys, time
import torch
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer
dev = torch.device("cuda:0")
for b in [8192, 8191, 8190, 8128, 8000, 8200, 8256, 4096, 4100, 6000, 12288, 10000, 16384, 16000]:
    cfg = TwoTowerConfig(n_users=2_000_000, n_items=1_000_000, embedding_dim=128, tower_dims=[256, 128], temperature=0.1,
                         l2_regularization=1e-6, learning_rate=0.001, optimizer="sgd", batch_size=b)
    tr = TwoTowerTrainer(cfg, dev, seed=1)
    batches = [tr.synthetic_batch(1, s) for s in range(8)]
    for s in range(30):
        tr.step(*batches[s % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 150
    for s in range(n):
        tr.step(*batches[s % 8])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"batch": b, "ms_per_step": round(ms, 4), "pairs_per_s": round(b / ms * 1e3), "ns_per_pair2": round(ms * 1e6 / (b * b / 8192), 2),
                      "composite": tr._cstep is not None}), flush=True)
    del tr
    torch.cuda.empty_cache()
