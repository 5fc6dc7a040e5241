"""Round 4: the row-sharded step on ONE rank (collectives forced: RCCL with no traffic) against the batch size."""
import json, os, sys, time
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig
from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
for b in [8192, 8200, 8000, 4096, 4100, 2048, 2000, 16384]:
    for neg in ("local", "global"):
        cfg = TwoTowerConfig(n_users=2_000_000, n_items=1_000_000, embedding_dim=128, tower_dims=[256, 128], temperature=0.1,
                             l2_regularization=1e-6, learning_rate=0.001, optimizer="sgd", batch_size=b)
        tr = ShardedTwoTowerTrainer(cfg, dev, seed=1, negatives=neg, force_collectives=True)
        batches = [tr.synthetic_batch(1, s) for s in range(8)]
        for s in range(20): tr.step(*batches[s % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 100
        for s in range(n): tr.step(*batches[s % 8])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        tr.check_ids()
        print(json.dumps({"batch": b, "negatives": neg, "ms_per_step": round(ms, 4), "pairs_per_s": round(b / ms * 1e3)}), flush=True)
        del tr
        torch.cuda.empty_cache()
dist.destroy_process_group()
