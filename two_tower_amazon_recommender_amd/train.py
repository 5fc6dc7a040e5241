"""``train-model`` / ``python -m two_tower_amazon_recommender_amd.train --config <yaml>`` — the training
entry point the reference declares (``/root/reference/pyproject.toml:67`` ``train-model =
"src.training.train:main"``; ``README.md:39`` ``python src/training/train.py --config ...``) but does not
ship.  Reads the ``model:`` block of the reference's YAML schema (``configs/data_config.yaml:54-71``) and the
parquet written by ``prepare_training_data.py:216-218``; runs every step on the HIP kernels (one GPU here;
under ``python -m torch.distributed.run --nproc-per-node N`` — or with ``--distributed`` — one process per GPU on the
row-sharded trainer of ``sharded.py``: every rank trains on its slice of the interactions, tables are sharded by
``id % N``, collectives go over RCCL).
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import time

import numpy as np
import torch

from . import config as cfgmod
from . import data as datamod
from .trainer import TwoTowerTrainer

log = logging.getLogger("train")


class EarlyStopping:
    """`model.training.patience` of the reference's schema (configs/data_config.yaml:65): stop after `patience`
    consecutive evaluations without an improvement of the validation loss by more than `min_delta`."""

    def __init__(self, patience: int, min_delta: float = 1e-6):
        self.patience, self.min_delta = int(patience), float(min_delta)
        self.best, self.bad = float("inf"), 0

    def update(self, value: float) -> bool:
        """Record one evaluation; returns True when training should stop."""
        if value < self.best - self.min_delta:
            self.best, self.bad = value, 0
        else:
            self.bad += 1
        return self.bad >= self.patience

    def state_dict(self) -> dict:
        return {"best": self.best, "bad": self.bad}

    def load_state_dict(self, sd: dict):
        self.best, self.bad = float(sd["best"]), int(sd["bad"])


def parse(argv=None):
    ap = argparse.ArgumentParser(description="Train the two-tower retrieval model on MI355X (HIP kernels).")
    ap.add_argument("--config", required=True, help="YAML with a `model:` block (configs/data_config.yaml schema)")
    ap.add_argument("--data", default="data/processed/combined_interactions.parquet")
    ap.add_argument("--synthetic", type=int, default=0, metavar="N", help="train on N synthetic interactions instead of --data")
    ap.add_argument("--synthetic-users", type=int, default=10_000)
    ap.add_argument("--synthetic-items", type=int, default=10_000)
    ap.add_argument("--optimizer", default="adagrad", choices=["sgd", "adagrad"])
    ap.add_argument("--category-buckets", type=int, default=0, metavar="N",
                    help="add the hashed category feature: the pair's category (column category / main_category / "
                         "category_encoded) hashed into N buckets, its embedding summed into the item tower input")
    ap.add_argument("--correct-sampling-bias", action="store_true",
                    help="pass every candidate's empirical frequency as candidate_sampling_probability (the logQ correction "
                         "of tfrs.tasks.Retrieval): in-batch negatives otherwise push popular items down")
    ap.add_argument("--scorer-precision", default="f32", choices=["f32", "bf16x3"],
                    help="matrix products of the scorer + softmax loss: exact f32 (default) or f32-emulated split-bf16 on the "
                         "bf16 matrix cores (scorer dim 128 / 256; same 1e-4 parity bars, ~2x faster scorer)")
    ap.add_argument("--epochs", type=int, default=None, help="override model.training.epochs")
    ap.add_argument("--batch-size", type=int, default=None, help="override model.training.batch_size")
    ap.add_argument("--val-fraction", type=float, default=0.1)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--save", default=None, help="write a checkpoint (torch.save of tensors) here at the end "
                                                 "(distributed: one file per rank, <path>.rank<r>of<N>)")
    ap.add_argument("--resume", default=None, help="continue from a checkpoint written by --save (distributed: the "
                                                   "<path>.rank<r>of<N> files of the same world size)")
    ap.add_argument("--distributed", action="store_true",
                    help="use the row-sharded multi-GPU trainer (implied when WORLD_SIZE > 1); batch_size is per rank")
    ap.add_argument("--negatives", default="local", choices=["local", "global"],
                    help="distributed: in-batch negatives of the rank's own batch (what tfrs.tasks.Retrieval sees under a "
                         "data-parallel strategy) or of the all-gathered global batch")
    return ap.parse_args(argv)


def main(argv=None) -> int:
    args = parse(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    doc = cfgmod.load_yaml(args.config)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    distributed = args.distributed or world > 1
    if distributed:
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world > 1 or "LOCAL_RANK" in os.environ:
            args.device = f"cuda:{local_rank}"
        torch.cuda.set_device(torch.device(args.device))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511")):
            os.environ.setdefault(k, v)
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device(args.device))
    if args.synthetic:
        from . import ops
        dev = torch.device(args.device)
        u = torch.empty(args.synthetic, dtype=torch.int64, device=dev)
        i = torch.empty(args.synthetic, dtype=torch.int64, device=dev)
        ops.fill_ids_(u, args.seed, 3, args.synthetic_users, "Z")
        ops.fill_ids_(i, args.seed, 4, args.synthetic_items, "Z")
        user_idx, item_idx = u.cpu().numpy(), i.cpu().numpy()
        n_users, n_items = args.synthetic_users, args.synthetic_items
        cat = None
        if args.category_buckets:
            c = torch.empty(args.synthetic, dtype=torch.int64, device=dev)
            ops.fill_ids_(c, args.seed, 6, args.category_buckets, "Z")
            cat = c.cpu().numpy()
    else:
        user_idx, item_idx = datamod.read_interactions(args.data)
        n_users, n_items = int(user_idx.max()) + 1, int(item_idx.max()) + 1
        cat = None
        if args.category_buckets:
            cv = datamod.read_category_values(args.data)
            if cv is None:
                raise SystemExit(f"--category-buckets: {args.data} has none of the columns {datamod.CATEGORY_COLUMNS}")
            cat = datamod.category_buckets(cv[0], cv[1], args.category_buckets, torch.device(args.device))
    cfg, loop = cfgmod.model_config_from_dict(doc, n_users, n_items, optimizer=args.optimizer)
    cfg.n_category_buckets = args.category_buckets
    cfg.scorer_precision = args.scorer_precision
    if args.batch_size:
        cfg.batch_size = args.batch_size
    epochs = args.epochs if args.epochs is not None else loop["epochs"]
    n = len(user_idx)
    n_val = int(n * args.val_fraction)
    rng = np.random.default_rng(args.seed)
    perm = rng.permutation(n)
    tr_idx, va_idx = perm[n_val:], perm[:n_val]
    log.info("users %d items %d interactions %d (train %d, val %d); batch %d; optimizer %s", n_users, n_items, n,
             len(tr_idx), len(va_idx), cfg.batch_size, cfg.optimizer)
    if distributed:
        # every rank computed the same split; it trains on every world-th pair, cut so all ranks run the same number
        # of (collective) steps
        from .sharded import ShardedTwoTowerTrainer
    # whole (global) batches only — the kernels' buffers are sized for one batch size — then every world-th pair
    per = cfg.batch_size * world
    tr_idx = tr_idx[:len(tr_idx) // per * per][rank::world]
    va_idx = va_idx[:len(va_idx) // per * per][rank::world]
    if len(tr_idx) < cfg.batch_size:
        raise SystemExit(f"only {len(tr_idx)} training interactions per rank for batch_size {cfg.batch_size}")
    if distributed:
        trainer = ShardedTwoTowerTrainer(cfg, args.device, seed=args.seed, negatives=args.negatives)
    else:
        trainer = TwoTowerTrainer(cfg, args.device, seed=args.seed)

    def total(x: torch.Tensor) -> float:        # sum over ranks of a device scalar
        if distributed:
            dist.all_reduce(x)
        return x.item()
    train_it = datamod.BatchIterator(user_idx[tr_idx], item_idx[tr_idx], cfg.batch_size, trainer.dev, args.seed,
                                     category_bucket=None if cat is None else cat[tr_idx])
    val_it = datamod.BatchIterator(user_idx[va_idx], item_idx[va_idx], cfg.batch_size, trainer.dev, args.seed, shuffle=False,
                                   category_bucket=None if cat is None else cat[va_idx])

    item_prob = None
    if args.correct_sampling_bias:          # P(item j is drawn as an in-batch candidate) = its share of the training pairs
        counts = torch.from_numpy(np.bincount(item_idx[tr_idx], minlength=n_items).astype(np.float64)).to(trainer.dev)
        if distributed:                     # the candidates' frequencies over ALL ranks' training pairs
            dist.all_reduce(counts)
        item_prob = (counts / counts.sum()).to(torch.float32)

    def kw(batch):
        k = {"category_ids": batch[2]} if len(batch) == 3 else {}
        if item_prob is not None:
            k["candidate_sampling_probability"] = item_prob[batch[1]]
        return k
    stopper, history, first_epoch = EarlyStopping(loop["patience"]), [], 0
    if args.resume:
        path = f"{args.resume}.rank{rank}of{world}" if distributed else args.resume
        ck = torch.load(path, map_location=trainer.dev, weights_only=True)       # plain tensors / numbers only
        trainer.load_state_dict(ck)
        first_epoch = int(ck.get("epoch", 0))
        if "early_stopping" in ck:
            stopper.load_state_dict(ck["early_stopping"])
        log.info("resumed from %s (epoch %d, step %d)", path, first_epoch, trainer.step_index)
    for epoch in range(first_epoch, epochs):
        t0 = time.perf_counter()
        tot = torch.zeros((), device=trainer.dev, dtype=torch.float64)
        for batch in train_it:
            tot.add_(trainer.step(batch[0], batch[1], **kw(batch)).view(()))      # (one mixed-precision add: f64 += f32)
        torch.cuda.synchronize()
        trainer.check_ids()
        dt = time.perf_counter() - t0
        rec = {"epoch": epoch + 1, "train_loss_per_pair": total(tot) / (len(train_it) * cfg.batch_size * world),
               "pairs_per_sec": len(train_it) * cfg.batch_size * world / dt}
        if len(val_it) and (epoch + 1) % loop["validation_freq"] == 0:
            vt = torch.zeros((), device=trainer.dev, dtype=torch.float64)
            for batch in val_it:
                vt.add_(trainer.evaluate(batch[0], batch[1], **kw(batch)).view(()))
            rec["val_loss_per_pair"] = total(vt) / (len(val_it) * cfg.batch_size * world)
            stop = stopper.update(rec["val_loss_per_pair"])
        else:
            stop = False
        history.append(rec)
        log.info(json.dumps(rec))
        last_epoch = epoch + 1
        if stop:                                         # early stopping (configs/data_config.yaml:65)
            log.info("early stop: no validation improvement for %d evaluations", stopper.bad)
            break
    final = {"history": history}
    if loop["top_k_eval"] and len(val_it):
        # retrieval quality on the held-out pairs against the WHOLE item corpus (configs/data_config.yaml:71 top_k_eval)
        from .metrics import FactorizedTopK
        metric = FactorizedTopK(ks=tuple(loop["top_k_eval"]), temperature=cfg.temperature)
        item_cat = None
        if cat is not None:             # an item's category = the bucket of its first interaction (items never seen: bucket 0)
            item_cat_np = np.zeros(n_items, dtype=np.int64)
            first = np.unique(item_idx, return_index=True)
            item_cat_np[first[0]] = cat[first[1]]
            item_cat = torch.from_numpy(item_cat_np).to(trainer.dev)
        corpus = trainer.item_corpus_embeddings(item_cat)
        for batch in val_it:
            trainer.evaluate_topk(batch[0], batch[1], metric, corpus)
        if distributed and metric._n:           # every rank ranked its own held-out pairs: sum the tallies
            dist.all_reduce(metric._hits); dist.all_reduce(metric._dcg)
            metric._n *= world
        final["val_metrics"] = {k: float(v) for k, v in metric.result().items()}
        log.info(json.dumps(final["val_metrics"]))
    if args.save:
        path = f"{args.save}.rank{rank}of{world}" if distributed else args.save
        ck = trainer.state_dict()
        ck.update(epoch=last_epoch if history else first_epoch, early_stopping=stopper.state_dict())
        torch.save(ck, path)
        log.info("saved checkpoint to %s", path)
    if rank == 0:
        print(json.dumps(final))
    if distributed:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
