"""N>1 leg of bench.py: one rank per GPU (torch.distributed.run), tables row-sharded, RCCL over xGMI.

Workloads (``--config``):
  cfg3 (default)  the weak-scaled family of the N=1 line (BASELINE configs[2] per GPU): per-GPU batch 8192, 5M users
                  and 10M items PER GPU, emb_dim 128, towers 128->256->128, SGD          -> "scaling": "weak"
  cfg4            BASELINE configs[3]: 100M-row item table (5M users) row-sharded over the N GPUs, emb_dim 128,
                  GLOBAL batch 16384 (16384/N per GPU), towers 128->256->128, SGD       -> "scaling": "strong"
  cfg5            BASELINE configs[4]: 54M users x 48M items, emb_dim 256, GLOBAL batch 32768, towers
                  256->512->256, fused sparse Adagrad, 30-bucket hashed category feature -> "scaling": "strong"
In-batch negatives: cfg4 / cfg5 (BASELINE's fixed global problems) score every rank's queries against the GLOBAL batch
(candidates all-gathered, dC reduce-scattered: the loss equals the single-device loss on the global batch).
The weak-scaled cfg3 family scores each rank's queries against its OWN batch (``local``: what a data-parallel Keras replica
of the reference's stack computes) - since r04: "weak" means the work per GPU is fixed as N grows, and with global negatives
every GPU's scorer would do N times the N = 1 work (B_local x N B_local logits; through r03 that was this line, its numbers
are the ``other_negatives`` object now).  ``--negatives global|local`` overrides; the mode is named in ``config.workload``
and the OTHER mode is timed in the same run (``other_negatives``).
value = global batch * steps / max-over-ranks time.  The line carries the scorer's ``roofline`` on the per-GPU slab
(B_local x B_global) from live dispatch timestamps (hipExtLaunchKernelGGL event pair, r04) on rank 0, and per-collective stream time from an untimed detail pass.
"""
import json
import os
import time

import torch
import torch.distributed as dist

MFMA_F32_PEAK_TFLOPS = 157.3

# name: (n_users, n_items, emb_dim, tower_dims, global_batch or None (= 8192 per GPU), optimizer, category buckets)
DIST_CONFIGS = {
    "cfg3": (5_000_000, 10_000_000, 128, [256, 128], None, "sgd", 0),        # rows and batch are PER GPU
    "cfg4": (5_000_000, 100_000_000, 128, [256, 128], 16384, "sgd", 0),
    "cfg5": (54_000_000, 48_000_000, 256, [512, 256], 32768, "adagrad", 30),
}


def _timed(fn, stream_events):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = fn()
    b.record()
    stream_events.append((a, b))
    return out


def collective_detail(trainer, batches, steps):
    """Per-collective stream time (us, mean over `steps` steps) of rank 0: every torch.distributed call of a step is
    bracketed with events on the stream it is enqueued on (the asynchronous gradient all-to-all is issued inline for
    this pass, so its bracket is the collective alone).  Untimed: runs after the measured region."""
    names = ("all_to_all_single", "all_gather_into_tensor", "reduce_scatter_tensor", "all_reduce")
    rec = {n: [] for n in names}
    orig = {n: getattr(dist, n) for n in names}

    def wrap(n):
        def f(*a, **k):
            k.pop("async_op", None)
            _timed(lambda: orig[n](*a, **k), rec[n])
            return None
        return f
    for n in names:
        setattr(dist, n, wrap(n))
    trainer.emb._wait = lambda w: None            # (instance attribute: shadows the class's static method)
    try:
        for s in range(steps):
            trainer.step(*batches[s][:2], **batches[s][2])
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(dist, n, orig[n])
        del trainer.emb._wait
    out = {}
    for n, ev in rec.items():
        if ev:
            us = [a.elapsed_time(b) * 1e3 for a, b in ev]
            out[n] = {"calls_per_step": len(us) / steps, "us_per_call": sum(us) / len(us), "us_per_step": sum(us) / steps}
    return out


def run_distributed(args, rank, world, dev):
    from two_tower_amazon_recommender_amd import _lib
    from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer
    from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "RANK" not in os.environ:        # TT_FORCE_DIST=1 without a launcher: a one-rank group
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
    dist.init_process_group("nccl", device_id=dev)
    name = args.config if args.config in DIST_CONFIGS else "cfg3"
    n_users, n_items, dim, tower_dims, gbatch, opt, buckets = DIST_CONFIGS[name]
    negatives = args.negatives or ("local" if name == "cfg3" else "global")
    if name == "cfg3":                  # weak scaling: fixed work per GPU
        n_users, n_items = int(os.environ.get("TT_USERS_PER_GPU", n_users)) * world, int(os.environ.get("TT_ITEMS_PER_GPU", n_items)) * world
        batch, scaling = 8192, "weak"
        opt = args.optimizer or "sgd"
    else:                               # BASELINE's 8-GPU configurations: fixed global problem
        if gbatch % world:
            raise SystemExit(f"{name}: global batch {gbatch} is not divisible by {world} ranks")
        batch, scaling = gbatch // world, "strong"
    seed = 1000 + int(name[3:])
    cfg = TwoTowerConfig(n_users=n_users, n_items=n_items, embedding_dim=dim, tower_dims=tower_dims, temperature=0.1,
                         l2_regularization=1e-6, learning_rate=0.001, optimizer=opt, batch_size=batch,
                         n_category_buckets=buckets)
    # TT_FORCE_COLLECTIVES=1: issue the RCCL calls even on one rank (their launch cost without any xGMI traffic)
    trainer = ShardedTwoTowerTrainer(cfg, dev, seed=seed, negatives=negatives,
                                     capacity_factor=float(os.environ.get("TT_CAPACITY_FACTOR", 2.0)),
                                     force_collectives=bool(os.environ.get("TT_FORCE_COLLECTIVES")),
                                     sync_ops_inline=os.environ.get("TT_SYNC_OPS_INLINE", "1") == "1")
    total = args.warmup + args.steps
    uids = torch.empty(total, batch, dtype=torch.int64, device=dev)
    iids = torch.empty(total, batch, dtype=torch.int64, device=dev)
    cids = torch.empty(total, batch, dtype=torch.int64, device=dev) if buckets else None
    for s in range(total):
        trainer.synthetic_batch(seed, s, args.ids, out=(uids[s], iids[s]))
        if cids is not None:
            trainer.synthetic_categories(seed, s, out=cids[s])
    # TT_PREFETCH=1: the next batch's ids are handed to step(), which routes them on a side stream beside this step's
    # scorer; TT_PREFETCH=2 also issues their id all-to-all early (off by default: see DESIGN.md §6).
    pf = os.environ.get("TT_PREFETCH", "0")
    prefetch, pf_exchange = pf in ("1", "2"), pf == "2"
    batches = []
    for s in range(total):
        kw = {} if cids is None else {"category_ids": cids[s]}
        if prefetch and s + 1 < total:
            kw.update(next_ids=(uids[s + 1], iids[s + 1]), prefetch_exchange=pf_exchange)
        batches.append((uids[s], iids[s], kw))

    def step(s):
        return trainer.step(batches[s][0], batches[s][1], **batches[s][2])
    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    trainer.check_ids()
    stride = 4 if args.steps >= 40 else 1
    if rank == 0:
        _lib.profile_set_stride(stride)
        _lib.profile_enable("score_fused", capacity=2 * args.steps + 8)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.warmup, total):
        step(s)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    loss = trainer.loss.double().clone()
    dist.all_reduce(loss, op=dist.ReduceOp.SUM)
    trainer.check_ids()
    fused = []
    if rank == 0:
        fused = _lib.profile_read("score_fused", 2 * args.steps + 8)[0]
        _lib.profile_enable("")
    detail_steps = min(args.steps, 20)
    coll = collective_detail(trainer, batches[total - detail_steps:], detail_steps) if trainer.collectives else {}
    trainer.check_ids()
    # second, separately labelled region: the same steps with the OTHER negatives mode (global <-> local).  With the
    # weak-scaled family and global negatives the scorer's work per GPU grows with N (B_local x N*B_local logits); the
    # local line shows the exchange + towers + optimizer scaling alone.
    other = "local" if negatives == "global" else "global"
    alt = None
    if trainer.collectives and world > 1:
        trainer.set_negatives(other)
        alt_steps = min(args.steps, 50)
        for s in range(min(args.warmup, 5)):
            step(s)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for s in range(total - alt_steps, total):
            step(s)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dta = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=dev)
        dist.all_reduce(dta, op=dist.ReduceOp.MAX)
        alt = (alt_steps, dta.item())
        trainer.set_negatives(negatives)
        trainer.check_ids()
    if rank == 0:
        sec = dt.item()
        sd = tower_dims[-1]
        nc = batch * world if (negatives == "global" and trainer.collectives) else batch
        roofline = None
        if fused:
            t = sum(fused) / len(fused) * 1e-3
            flops = 4.0 * batch * nc * sd                  # pass 1 on this GPU's slab: logits + dq (algorithmic = executed)
            a = flops / t / 1e12
            roofline = {"bound": "mfma", "kernel": f"score_kernel<{sd},FUSED_S> on rank 0's slab [{batch} queries x {nc} candidates] "
                                                   "(loss + dq pass; algorithmic 4*Bq*Bc*D per launch)",
                        "achieved": a, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": a / MFMA_F32_PEAK_TFLOPS,
                        "traffic": None, "avg_launch_us": t * 1e6, "samples": len(fused), "dtype": "f32-input MFMA"}
        tower = f"{dim}->" + "->".join(map(str, tower_dims))
        # How to read the curve: per-GPU work at this N against N = 1.  With GLOBAL negatives every GPU scores its B_local
        # queries against all N*B_local candidates, so in the weak-scaled family (cfg3) the scorer's FLOPs per GPU grow N-fold
        # by construction - a flat pairs/s-per-GPU curve would be super-linear.  ms_expected_from_n1 = the committed N = 1
        # kernel times (profiles/), the scorer part rescaled by that factor, + this run's measured stream time of every
        # collective of a step: what the step would take if nothing but the model's terms changed.
        work_model = None
        try:
            prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            ref = next(f for f in ("r04_n1_reference.json", "r03_n1_reference.json") if os.path.exists(os.path.join(prof_dir, f)))
            n1 = json.load(open(os.path.join(prof_dir, ref)))[name]
            factor = (nc / batch) * (batch / n1["batch_per_gpu"]) ** 2      # Bq*Bc*D against the N = 1 launch's
            other = n1["ms_per_step"] - n1["scorer_ms"]
            if name != "cfg3":
                other *= batch / n1["batch_per_gpu"]                        # towers / lookup / optimizer scale with the local batch
            coll_ms = sum(v["us_per_call"] * v["calls_per_step"] for v in coll.values()) * 1e-3 if coll else 0.0
            work_model = {"scorer_flops_per_gpu_vs_n1": factor, "n1_reference": n1,
                          "collectives_ms_per_step_measured": coll_ms,
                          "ms_expected_from_n1": n1["scorer_ms"] * factor + other + coll_ms,
                          "note": "a model: N = 1 kernel times rescaled + measured per-collective stream time; exposed waits "
                                  "between ranks are what the difference to ms_per_step shows"}
        except (OSError, KeyError, ValueError, ZeroDivisionError, TypeError, StopIteration):
            pass
        out = {
            "metric": "user-item pairs/sec (train step) + embedding-gather HBM GB/s, 1/2/4/8 MI355X",
            "value": world * batch * args.steps / sec, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": sec / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{name}{' per GPU (weak-scaled family of the N=1 line)' if name == 'cfg3' else ''}, row-sharded: "
                                    f"{cfg.n_users} users x {cfg.n_items} items over {world} GPUs (owner = id % {world}), "
                                    f"emb_dim {dim}, towers {tower}, batch {batch}/GPU (global {world * batch}), in-batch "
                                    f"negatives {negatives.upper()}"
                                    + (f" (every GPU scores its {batch} queries against all {world * batch} candidates: scorer "
                                       f"work per GPU grows with N)" if negatives == "global" and name == "cfg3" and world > 1 else "")
                                    + f", {opt} lr 1e-3, ids {args.ids}"
                                    + (f", + {buckets}-bucket hashed category feature" if buckets else "")),
                       "global_batch": world * batch, "parallelism": f"dp{world} + row-sharded tables (all-to-all)",
                       "negatives": negatives},
            "roofline": roofline, "cpu_baseline": None,
            "collectives": coll or None,
            "work_model": work_model,
            "other_negatives": None if alt is None else {
                "negatives": other, "value": world * batch * alt[0] / alt[1], "unit": "pairs/s", "ms_per_step": alt[1] / alt[0] * 1e3,
                "steps": alt[0], "note": "same run, same steps, in-batch negatives switched; NOT the headline value"},
            "timing_note": f"dispatch timestamps inside the timed region: score_fused on rank 0, every {stride}th step; "
                           f"collectives: stream time per call, untimed detail pass of {detail_steps} steps",
            "loss_per_pair": loss.item() / (world * batch),
        }
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()
