"""Host enqueue time vs GPU time per step (plain trainer, sharded world=1, sharded with forced one-rank RCCL)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer
from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer

dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
def cfg():
    return TwoTowerConfig(n_users=5_000_000, n_items=10_000_000, embedding_dim=128, tower_dims=[256, 128], temperature=0.1,
                          l2_regularization=1e-6, learning_rate=0.001, optimizer="sgd", batch_size=8192)
for name, mk in (("plain", lambda: TwoTowerTrainer(cfg(), dev, seed=3)),
                 ("sharded w=1", lambda: ShardedTwoTowerTrainer(cfg(), dev, seed=3)),
                 ("sharded w=1 + RCCL calls", lambda: ShardedTwoTowerTrainer(cfg(), dev, seed=3, force_collectives=True))):
    tr = mk()
    pre = name != "plain" and os.environ.get("TT_PREFETCH", "1") == "1"
    n = 300
    ids = [tr.synthetic_batch(3, s, "U") for s in range(n + 20)]
    for s in range(20):
        tr.step(*ids[s], **({"next_ids": ids[s + 1]} if pre and s + 1 < len(ids) else {}))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(20, n + 20):
        tr.step(*ids[s], **({"next_ids": ids[s + 1]} if pre and s + 1 < len(ids) else {}))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:28s} host enqueue {1e3*(t1-t0)/n:.3f} ms/step   total {1e3*(t2-t0)/n:.3f} ms/step", flush=True)
    del tr
    torch.cuda.empty_cache()
dist.destroy_process_group()
