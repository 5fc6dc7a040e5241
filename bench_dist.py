"""N>1 leg of bench.py: one rank per GPU (torch.distributed.run), RCCL over xGMI.

Workload family (weak scaling, consistent with the N=1 line = BASELINE cfg3): per-GPU batch 8192,
tables row-sharded with 5M users and 10M items PER GPU (N=8: 40M x 80M rows; BASELINE cfg4's 100M-row item
table is `--items-per-gpu 12500000`), emb_dim 128, towers 128->256->128, in-batch negatives per rank
(`--negatives local`, what tfrs.tasks.Retrieval sees under a data-parallel strategy) or all-gathered
(`--negatives global`).  value = world * batch * steps / max-over-ranks time.
"""
import json
import os
import time

import torch
import torch.distributed as dist


def run_distributed(args, rank, world, dev):
    from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer
    from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world == 1:                      # TT_FORCE_DIST=1 without a launcher: a one-rank group
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
    dist.init_process_group("nccl", device_id=dev)
    negatives = os.environ.get("TT_NEGATIVES", "local")
    users_per_gpu = int(os.environ.get("TT_USERS_PER_GPU", 5_000_000))
    items_per_gpu = int(os.environ.get("TT_ITEMS_PER_GPU", 10_000_000))
    batch, dim, tower_dims = 8192, 128, [256, 128]
    seed = 1003
    cfg = TwoTowerConfig(n_users=users_per_gpu * world, n_items=items_per_gpu * world, embedding_dim=dim,
                         tower_dims=tower_dims, temperature=0.1, l2_regularization=1e-6, learning_rate=0.001,
                         optimizer=args.optimizer, batch_size=batch)
    # TT_FORCE_COLLECTIVES=1: issue the RCCL calls even on one rank (their launch cost without any xGMI traffic)
    trainer = ShardedTwoTowerTrainer(cfg, dev, seed=seed, negatives=negatives,
                                     capacity_factor=float(os.environ.get("TT_CAPACITY_FACTOR", 2.0)),
                                     force_collectives=bool(os.environ.get("TT_FORCE_COLLECTIVES")),
                                     sync_ops_inline=os.environ.get("TT_SYNC_OPS_INLINE", "1") == "1")
    total = args.warmup + args.steps
    uids = torch.empty(total, batch, dtype=torch.int64, device=dev)
    iids = torch.empty(total, batch, dtype=torch.int64, device=dev)
    for s in range(total):
        trainer.synthetic_batch(seed, s, args.ids, out=(uids[s], iids[s]))
    # TT_PREFETCH=1: the next batch's ids are handed to step(), which routes them on a side stream beside this step's
    # scorer; TT_PREFETCH=2 also issues their id all-to-all early.  Measured on one rank (forced RCCL calls) both cost
    # more than they hide (0.888 -> 0.896 / 0.907 ms: every cross-stream hand-off is a 4-7 us bubble in the GPU queue
    # and the early collective disturbs the scorer), so neither is the default until measured on a multi-GPU node.
    pf = os.environ.get("TT_PREFETCH", "0")
    prefetch, pf_exchange = pf in ("1", "2"), pf == "2"
    batches = [(uids[s], iids[s]) for s in range(total)] + [None]
    for s in range(args.warmup):
        trainer.step(*batches[s], next_ids=batches[s + 1] if prefetch else None, prefetch_exchange=pf_exchange)
    torch.cuda.synchronize()
    trainer.check_ids()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.warmup, total):
        trainer.step(*batches[s], next_ids=batches[s + 1] if prefetch else None, prefetch_exchange=pf_exchange)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    loss = trainer.loss.double().clone()
    dist.all_reduce(loss, op=dist.ReduceOp.SUM)
    trainer.check_ids()
    if rank == 0:
        sec = dt.item()
        out = {
            "metric": "user-item pairs/sec (train step) + embedding-gather HBM GB/s, 1/2/4/8 MI355X",
            "value": world * batch * args.steps / sec, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": sec / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg3 per GPU, row-sharded: {cfg.n_users} users x {cfg.n_items} items over {world} GPUs "
                                   f"(owner = id % {world}), emb_dim {dim}, towers {dim}->256->128, batch {batch}/GPU, "
                                   f"in-batch negatives {negatives}, {args.optimizer} lr 1e-3, ids {args.ids}",
                       "global_batch": world * batch, "parallelism": f"dp{world} + row-sharded tables (all-to-all)"},
            "roofline": None, "cpu_baseline": None,
            "loss_per_pair": loss.item() / (world * batch),
        }
        print(json.dumps(out))
    dist.destroy_process_group()
