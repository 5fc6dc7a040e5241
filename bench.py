"""bench.py — pairs/sec of the two-tower train step (fwd + bwd + optimizers) on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1 is launched by the driver through
torch.distributed.run, one rank per GPU).  Rank 0 prints ONE JSON line.

Workload at N=1: BASELINE.json configs[2] (the configuration north_star quotes its targets on,
"batch 8192/dim 128"): 10M items x 5M users, emb_dim 128, towers 128->256(ReLU)->128, B = 8192,
temperature 0.1, SGD lr 0.001, uniform ids, synthetic counter-based data.
`--config cfg1|cfg2|cfg3` selects another single-GPU configuration.

Inputs (tables, weights, id batches) are resident in HBM before the timed region.  A step is one
pass of the hot path over one batch.  The `roofline` object is measured live, inside the timed
steps: sampled launches of the dominant kernel go out through hipExtLaunchKernelGGL with a HIP event
pair that the runtime fills with the dispatch's own begin / end timestamps on its stream
(tt_profile_enable; r04 - the figures rocprofv3 --kernel-trace reports, no barrier packets; through
r03 hipEventRecord brackets, ~3 us longer per launch); `cpu_baseline` times the torch-CPU
restatement (oracle/torch_cpu.py, kind "port") on a bounded sample of the same batches, rank 0, N=1.

Extra, separately labelled lines for profiles/ (never the headline): `--config ref` = the only
configuration the reference writes down (/root/reference/configs/data_config.yaml:54-71:
embedding_dim 128, towers [512, 256, 128], batch 1024, dropout 0.1; table sizes are cfg3's - the
reference names none; Adagrad); `--engine ops` = the same step through the boundary north_star
names: tasks.Retrieval + torch.ops.twotower.* + torch.autograd instead of the explicit trainer.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (n_users, n_items, emb_dim, tower_dims, batch)           SURVEY.md §8d
    "cfg1": (10_000, 10_000, 32, [32], 256),
    "cfg2": (1_000_000, 1_000_000, 64, [64], 4096),
    "cfg3": (5_000_000, 10_000_000, 128, [256, 128], 8192),
    # BASELINE configs[3] / [4] are 8-GPU configurations (`--gpus N --config cfg4|cfg5`: bench_dist.py); their tables
    # also fit ONE MI355X (288 GB), so N=1 measures them un-sharded
    "cfg4": (5_000_000, 100_000_000, 128, [256, 128], 16384),
    "cfg5": (54_000_000, 48_000_000, 256, [512, 256], 32768),
    # the reference's own `model:` block (configs/data_config.yaml:54-71); it names no table sizes: cfg3's
    "ref": (5_000_000, 10_000_000, 128, [512, 256, 128], 1024),
}
REF_DROPOUT = 0.1                   # configs/data_config.yaml:58
CATEGORY_BUCKETS = {"cfg5": 30}     # BASELINE configs[4]: "30 categories as hash features" (summed into the item tower input)
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32-input MFMA dense peak
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--optimizer", default=None, choices=["sgd", "adagrad"],
                    help="default: sgd (cfg5: adagrad — BASELINE configs[4] names the fused sparse Adagrad)")
    ap.add_argument("--ids", default="U", choices=["U", "Z"], help="uniform / power-law id batches")
    ap.add_argument("--engine", default="trainer", choices=["trainer", "ops"],
                    help="trainer: the explicit engine behind tt_train_step_f32 (headline); ops: tasks.Retrieval + "
                         "torch.ops.twotower.* + autograd (the boundary north_star names; an extra line, never the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    ap.add_argument("--pre-spin-ms", type=float, default=250.0,
                    help="before the W warm-up steps, keep the GPU busy this long with forward-only passes over the first "
                         "batch (no parameter is touched): a fresh process starts at idle clocks and cold TLBs, and a "
                         "25-step run otherwise measures the ramp (BENCH_r01: FUSED kernel 290 us vs 269 us in steady state)")
    ap.add_argument("--negatives", default=None, choices=["global", "local"],
                    help="N>1: in-batch negatives over the all-gathered GLOBAL batch (the loss of the single device on the global "
                         "batch; default for cfg4 / cfg5, BASELINE's fixed global problems) or over each rank's own batch (what a "
                         "data-parallel Keras replica computes; default for the weak-scaled cfg3 family: per-GPU work fixed as N "
                         "grows).  The other mode is timed in the same run and reported as `other_negatives`")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` from a bare shell: start the N ranks (one per GPU) through torch.distributed.run as a
    CHILD process — before this process has touched the GPU or imported torch — relay rank 0's JSON line and exit with
    the child's code.  (Never an exec: a process must not be replaced once it may have initialised the GPU.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        if line.lstrip().startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


def cpu_share() -> int:
    """Host threads this job may use: affinity, capped by the cgroup quota and by the GPU box's per-GPU
    CPU share (16)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


# rocprofv3 --pmc passes (scratch/prof_pmc.sh + pmc_summary.py) of the newest round that committed one
PMC_FILE = next((f for f in ("profiles/r04_pmc_cfg3_sgd.json", "profiles/r03_pmc_cfg3_sgd.json", "profiles/r02_pmc_cfg3_sgd.json")
                 if os.path.exists(os.path.join(ROOT, f))), "profiles/r02_pmc_cfg3_sgd.json")


# rocprofv3 --kernel-trace --stats of `bench.py --steps 200 --warmup 20` (scratch/prof.sh), newest committed round
KSTATS_FILE = next((f for f in ("profiles/r04_bench_cfg3_kernel_stats.csv", "profiles/r03_bench_cfg3_kernel_stats.csv",
                                "profiles/r02_bench_cfg3_kernel_stats.csv")
                    if os.path.exists(os.path.join(ROOT, f))), None)


def rocprof_avg_us(*needles):
    """Sum of the average durations (us) of the kernels whose name contains one of `needles`, from the COMMITTED rocprofv3
    kernel stats of the cfg3 / sgd step - not measured in this run; since r04 the live figures are the same dispatch
    timestamps, so this is a cross-check only.  None when the file or a kernel is missing."""
    try:
        import csv
        rows = list(csv.DictReader(open(os.path.join(ROOT, KSTATS_FILE))))
        tot = 0.0
        for n in needles:
            hit = [r for r in rows if n in r["Name"]]
            if not hit:
                return None
            tot += sum(float(r["AverageNs"]) for r in hit) * 1e-3
        return tot
    except (OSError, KeyError, ValueError, TypeError):
        return None


def rocprof_lookup_share_us():
    """K1's share of the layer-0 launches from the COMMITTED rocprofv3 A/B (lookup fused vs materialised input, same box,
    alternating: profiles/r04_lookup_share.json) - not measured in this run; the live figure beside it is a difference of
    two detail-pass means and moves by +-1 us from run to run."""
    try:
        return float(json.load(open(os.path.join(ROOT, "profiles", "r04_lookup_share.json")))["lookup_share_us"])
    except (OSError, KeyError, ValueError, TypeError):
        return None


def pmc_traffic(tag):
    """HBM bytes per launch of kernel `tag` from the COMMITTED rocprofv3 --pmc summary (separate passes,
    MI355X_MICROARCH.md §HBM corrections applied) — not measured in this run: the line says so in `traffic_source`."""
    try:
        return json.load(open(os.path.join(ROOT, PMC_FILE)))["kernels"][tag].get("hbm_bytes_per_launch")
    except (OSError, KeyError, ValueError):
        return None


def mean(xs):
    return sum(xs) / max(len(xs), 1)


def cpu_baseline(trainer, cfg, seed, args, batch):
    """torch-CPU restatement on the same model state and batches (bounded sample)."""
    from oracle.torch_cpu import TorchCpuTwoTower, time_cpu_steps      # bench's cpu_baseline leg only
    cores = cpu_share()
    torch.set_num_threads(cores)
    ut, it = trainer.user_table.cpu(), trainer.item_table.cpu()
    uw = [w.cpu().clone() for w in trainer.user_tower.w]; ub = [b.cpu().clone() for b in trainer.user_tower.b]
    iw = [w.cpu().clone() for w in trainer.item_tower.w]; ib = [b.cpu().clone() for b in trainer.item_tower.b]
    model = TorchCpuTwoTower(ut, it, uw, ub, iw, ib, temperature=cfg.temperature, l2=cfg.l2_regularization,
                             lr=cfg.learning_rate, optimizer=cfg.optimizer)
    batches = []
    for s in range(4):
        u, i = trainer.synthetic_batch(seed, 10_000 + s, args.ids)
        batches.append((u.cpu(), i.cpu()))
    sec, n = time_cpu_steps(model, batches, budget_s=args.cpu_budget_s)
    return {"value": batch / sec, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} train steps of the same workload (batch {batch}) with stock torch CPU ops, "
                      f"{sec * 1e3:.1f} ms/step; restatement, not reference code"}


def run_ops_engine(args, cfg, seed, dev):
    """The train step through the boundary `north_star` names (an EXTRA line for profiles/, never the headline):
    torch.ops.twotower.embedding_gather -> torch.ops.twotower.dense_fwd (autograd: dense_bwd) -> tasks.Retrieval
    (torch.ops.twotower.retrieval_loss, gradients saved for autograd) -> loss.backward() -> torch.ops.twotower.sparse_update_
    on the embedding-row gradients + a stock torch optimizer on the Dense parameters.  What a user of tfrs.tasks.Retrieval
    would write around the custom ops; every launch is its own FFI crossing, outputs are allocated per call."""
    from two_tower_amazon_recommender_amd import _lib
    from two_tower_amazon_recommender_amd.tasks import Retrieval
    from two_tower_amazon_recommender_amd.trainer import TwoTowerTrainer
    tr = TwoTowerTrainer(cfg, dev, seed=seed)              # (the same initial state and id batches as the trainer engine)
    batch, dims = cfg.batch_size, [cfg.embedding_dim] + list(cfg.user_dims)
    towers = []
    for tw in (tr.user_tower, tr.item_tower):
        towers.append(([w.detach().clone().requires_grad_(True) for w in tw.w], [b.detach().clone().requires_grad_(True) for b in tw.b]))
    params = [p for ws, bs in towers for p in ws + bs]
    dense_opt = (torch.optim.Adagrad(params, lr=cfg.learning_rate, initial_accumulator_value=cfg.adagrad_initial_accumulator,
                                     eps=cfg.adagrad_epsilon)
                 if cfg.optimizer == "adagrad" else torch.optim.SGD(params, lr=cfg.learning_rate))
    tables = (tr.user_table, tr.item_table)
    accums = (tr.user_accum, tr.item_accum) if cfg.optimizer == "adagrad" else (None, None)
    task = Retrieval(temperature=cfg.temperature)
    total = args.warmup + args.steps
    ids = [tr.synthetic_batch(seed, s, args.ids) for s in range(min(total, 64))]
    torch.cuda.synchronize()

    def tower(x, ws, bs):
        for l, (w, b) in enumerate(zip(ws, bs)):
            hidden = l < len(ws) - 1
            x = torch.ops.twotower.dense_fwd(x, w, b, hidden)
            if hidden and cfg.dropout_rate > 0:
                x = torch.nn.functional.dropout(x, cfg.dropout_rate)
        return x

    def step(s):
        u, i = ids[s % len(ids)]
        embs = [torch.ops.twotower.embedding_gather(tables[t], (u, i)[t]).requires_grad_(True) for t in range(2)]
        q, c = tower(embs[0], *towers[0]), tower(embs[1], *towers[1])
        loss = task(q, c, compute_metrics=False)
        dense_opt.zero_grad(set_to_none=True)
        loss.backward()
        dense_opt.step()                                   # (no l2 term here: the boundary leaves regularisation to the caller)
        for t in range(2):
            torch.ops.twotower.sparse_update_(tables[t], accums[t], embs[t].grad, (u, i)[t], cfg.optimizer, cfg.learning_rate,
                                              cfg.adagrad_epsilon)
        return loss

    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    _lib.profile_set_stride(4 if args.steps >= 8 else 1)
    _lib.profile_enable("score_fused", capacity=2 * args.steps + 8)
    t0 = time.perf_counter()
    for s in range(args.warmup, total):
        loss = step(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tf_ = mean(_lib.profile_read("score_fused", 2 * args.steps + 8)[0]) * 1e-3
    _lib.profile_set_stride(1)
    _lib.profile_enable("")
    sd = dims[-1]
    a = 4.0 * batch * batch * sd / tf_ / 1e12 if tf_ > 0 else None
    print(json.dumps({
        "metric": "user-item pairs/sec (train step) + embedding-gather HBM GB/s, 1/2/4/8 MI355X",
        "value": batch / (dt / args.steps), "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: {cfg.n_users} users x {cfg.n_items} items, emb_dim {cfg.embedding_dim}, towers "
                               f"{'->'.join(map(str, dims))}, batch {batch}, in-batch sampled softmax T={cfg.temperature}, "
                               f"{cfg.optimizer} lr {cfg.learning_rate}, dropout {cfg.dropout_rate}, ids {args.ids}",
                   "engine": "ops: tasks.Retrieval + torch.ops.twotower.* + torch.autograd (embedding_gather, dense_fwd / dense_bwd, "
                             "retrieval_loss, sparse_update_) + a stock torch optimizer on the Dense parameters",
                   "global_batch": batch, "parallelism": "single GPU"},
        "note": "EXTRA line (the boundary itself), not the headline: the headline engine is the explicit trainer behind tt_train_step_f32",
        "roofline": {"bound": "mfma", "kernel": f"score_kernel<{sd},FUSED_S> (loss + dq pass of torch.ops.twotower.retrieval_loss)",
                     "achieved": a, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": (a / MFMA_F32_PEAK_TFLOPS) if a else None,
                     "traffic": None, "avg_launch_us": tf_ * 1e6},
        "loss_per_pair": float(loss.item()) / batch}))


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:       # not under a launcher: become one
        raise SystemExit(self_launch(args))
    import torch                        # (imported here, not at module level: the self-launching parent never loads it)
    globals()["torch"] = torch
    if args.optimizer is None:
        args.optimizer = "adagrad" if args.config in ("cfg5", "ref") else "sgd"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from two_tower_amazon_recommender_amd import _lib, ops
    from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer

    if world > 1 or os.environ.get("TT_FORCE_DIST"):      # TT_FORCE_DIST=1: exercise the sharded path on one rank
        from bench_dist import run_distributed          # row-sharded tables + RCCL all-to-all
        return run_distributed(args, rank, world, dev)

    n_users, n_items, dim, tower_dims, batch = CONFIGS[args.config]
    seed = 1000 + (9 if args.config == "ref" else int(args.config[3:]))
    cfg = TwoTowerConfig(n_users=n_users, n_items=n_items, embedding_dim=dim, tower_dims=tower_dims,
                         temperature=0.1, l2_regularization=1e-6, learning_rate=0.001,
                         optimizer=args.optimizer, batch_size=batch,
                         dropout_rate=REF_DROPOUT if args.config == "ref" else 0.0,
                         n_category_buckets=CATEGORY_BUCKETS.get(args.config, 0))
    if args.engine == "ops":
        return run_ops_engine(args, cfg, seed, dev)
    trainer = TwoTowerTrainer(cfg, dev, seed=seed)
    total = args.warmup + args.steps
    uids = torch.empty(total, batch, dtype=torch.int64, device=dev)
    iids = torch.empty(total, batch, dtype=torch.int64, device=dev)
    cids = torch.empty(total, batch, dtype=torch.int64, device=dev) if cfg.n_category_buckets else None
    for s in range(total):
        trainer.synthetic_batch(seed, s, args.ids, out=(uids[s], iids[s]))
        if cids is not None:
            trainer.synthetic_categories(seed, s, out=cids[s])
    torch.cuda.synchronize()

    if args.graph:
        trainer.capture_graph()

    def step(s):
        if args.graph:
            return trainer.step_graph(uids[s], iids[s], None if cids is None else cids[s])
        if cids is None:
            return trainer.step(uids[s], iids[s])
        return trainer.step(uids[s], iids[s], category_ids=cids[s])
    if args.pre_spin_ms > 0:              # device warm-up, not training: forward-only, model state untouched
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < args.pre_spin_ms:
            for _ in range(8):
                if cids is None:
                    trainer.evaluate(uids[0], iids[0])
                else:
                    trainer.evaluate(uids[0], iids[0], category_ids=cids[0])
            torch.cuda.synchronize()
    for s in range(args.warmup):
        step(s)
    torch.cuda.synchronize()
    trainer.check_ids()

    # Timed region: only the DOMINANT kernel is timestamped (hipExtLaunchKernelGGL with an event pair: the dispatch's own begin /
    # end timestamps - no barrier packets, but a timestamped dispatch still costs the stream ~4 us: with all five kernel families
    # on at stride 4 the step was 0.5642 ms against 0.5584, r04 call 1).  The other kernels' durations come from an untimed detail
    # pass of the same steps right after it.
    all_tags = "score_fused,score_bwd,gather,sparse_plan,sparse_apply,optimizer,dense_fwd,dense_bwd".split(",")
    timed_tags = os.environ.get("TT_BENCH_TAGS", "score_fused")
    # the dominant kernel is bracketed on every 4th step only, whatever --steps is: with a bracket on EVERY launch the two
    # event records of consecutive launches sit back to back and the bracket itself reads ~20 us long (BENCH_r01: 290.6 us
    # at stride 1 in a 20-step run vs 269 us in rocprof and 271 us at stride 4)
    stride = 4 if args.steps >= 8 else 1
    _lib.profile_set_stride(stride)
    _lib.profile_enable(timed_tags, capacity=2 * args.steps + 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.warmup, total):
        step(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = {t: _lib.profile_read(t, 2 * args.steps + 8)[0] for t in timed_tags.split(",") if t in all_tags}
    detail_steps = min(args.steps, 100)
    rest = [t for t in all_tags if t not in prof]
    _lib.profile_set_stride(1)
    _lib.profile_enable(",".join(rest), capacity=2 * detail_steps + 8)
    for s in range(total - detail_steps, total):          # the same id batches again (state has moved on; shapes equal)
        step(s)
    torch.cuda.synchronize()
    for t in rest:
        prof[t] = _lib.profile_read(t, 2 * detail_steps + 8)[0]
    # K1's cost inside the first layer's GEMMs = (layer-0 fwd + bwd launch with the lookup fused) - (the same launches on a
    # materialised input), measured by a second untimed detail pass with the lookup un-fused (gather2 launch + acts[0])
    n_layers = len(tower_dims)
    lookup_us, unfused = None, {}
    lookup_fused = trainer.fuse_lookup          # (False by default when the first tower layer is 512 wide: trainer.py)
    if cfg.symmetric and not args.graph:
        trainer.fuse_lookup = trainer.fuse_optimizer = False      # every kernel as its own launch
        _lib.profile_enable("dense_fwd,dense_bwd,gather,sparse_apply,sparse_plan,dense_update", capacity=2 * n_layers * detail_steps + 8)
        for s in range(total - detail_steps, total):
            step(s)
        torch.cuda.synchronize()
        unfused = {t: _lib.profile_read(t, 2 * n_layers * detail_steps + 8)[0]
                   for t in ("dense_fwd", "dense_bwd", "gather", "sparse_apply", "sparse_plan", "dense_update")}
        trainer.fuse_lookup, trainer.fuse_optimizer = lookup_fused, True
        if lookup_fused:
            # (forward launches per step: n_layers, or ONE when the two-layer tower forward is fused - csrc/tower.hip - in both passes)
            fps = max(len(prof["dense_fwd"]) // detail_steps, 1)
            f_l0 = mean(prof["dense_fwd"][0::fps]) - mean(unfused["dense_fwd"][0::fps])
            b_l0 = mean(prof["dense_bwd"][n_layers - 1::n_layers]) - mean(unfused["dense_bwd"][n_layers - 1::n_layers])
            lookup_us = (f_l0 + b_l0) * 1e3
    # ---- second, separately labelled measurement: the same train step with the scorer's matrix products in the
    # f32-EMULATED bf16x3 precision (three-way bf16 split on the bf16 MFMA; same 1e-4 parity bars).  Never the headline:
    # `value` / `roofline` above are the exact-f32 run.
    alt = None
    sd = tower_dims[-1]
    if sd in (128, 256) and not args.graph:
        cfg.scorer_precision = "bf16x3"
        alt_steps = min(args.steps, 100)
        for s in range(min(args.warmup, 10)):
            step(s)
        torch.cuda.synchronize()
        _lib.profile_set_stride(4 if alt_steps >= 40 else 1)
        _lib.profile_enable("score_fused,score_bwd", capacity=2 * alt_steps + 8)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for s in range(total - alt_steps, total):
            step(s)
        torch.cuda.synchronize()
        dta = time.perf_counter() - ta
        af, ab = _lib.profile_read("score_fused", 2 * alt_steps + 8)[0], _lib.profile_read("score_bwd", 2 * alt_steps + 8)[0]
        _lib.profile_set_stride(1)
        cfg.scorer_precision = "f32"
        alt = (alt_steps, dta, mean(af) * 1e-3, mean(ab) * 1e-3, float(trainer.loss.item()))
    _lib.profile_enable("")
    steps_of = {t: (args.steps if t in timed_tags.split(",") else detail_steps) for t in all_tags}
    loss = float(trainer.loss.item())
    trainer.check_ids()

    ms_per_step = dt / args.steps * 1e3
    sd = tower_dims[-1]
    b2d = float(batch) * batch * sd
    if args.config == "ref":
        out_note_ref = ("the reference's own model block (configs/data_config.yaml:54-71): 3-layer towers run per-layer launches "
                        "(the fused two-layer tower forward does not apply), dropout 0.1 in the GEMM epilogues; an EXTRA line, not the headline")
    # The scorer+loss runs as two launches of one kernel template per step:
    #   score_kernel<D,FUSED_S>: loss + dq — algorithmic 4*B^2*D (fwd 2 + dq 2), executed 4*B^2*D; also writes the raw
    #                                        dot products [B][B] f32 to the workspace (4*B^2 bytes)
    #   score_kernel<D,BWD_S>  : dc        — algorithmic = executed 2*B^2*D: reads the dot products back (4*B^2 bytes)
    #                                        instead of recomputing them (r02: was 4*B^2*D executed)
    t_fused = mean(prof["score_fused"]) * 1e-3
    t_bwd = mean(prof["score_bwd"]) * 1e-3

    def roof(name, alg_flops, t, exec_flops=None):
        exec_flops = alg_flops if exec_flops is None else exec_flops
        if t <= 0:
            return {"bound": "mfma", "kernel": name, "achieved": None, "note": "no timestamp samples (graph replay)"}
        a = alg_flops / t / 1e12
        return {"bound": "mfma", "kernel": name, "achieved": a, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": a / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                "traffic_source": f"{PMC_FILE} (committed rocprofv3 --pmc passes; not re-measured in this run)",
                "traffic_note": "4*B^2 bytes each way (268 MB at B 8192) are the raw dot products pass 1 keeps for pass 2 - "
                                "24-27x the algorithmic q/c/gradient bytes, by design: it replaces a second GEMM1 (DESIGN.md section 4)",
                "avg_launch_us": t * 1e6, "dtype": "f32-input MFMA",
                "executed_tflops": exec_flops / t / 1e12, "executed_frac": exec_flops / t / 1e12 / MFMA_F32_PEAK_TFLOPS}

    r_fused = roof(f"score_kernel<{sd},FUSED_S> (loss + dq pass, keeps the [B][B] dot products for the dc pass; 1 launch/step; "
                   f"algorithmic = executed 4*B^2*D)", 4.0 * b2d, t_fused)
    r_bwd = roof(f"score_kernel<{sd},BWD_S> (dc pass from the stored dot products; 1 launch/step; algorithmic = executed "
                 f"2*B^2*D, + 4*B^2 bytes of HBM reads)", 2.0 * b2d, t_bwd)
    if args.config == "cfg3" and args.optimizer == "sgd":          # the configuration the PMC passes were collected on
        r_fused["traffic"], r_bwd["traffic"] = pmc_traffic("score_fused"), pmc_traffic("score_bwd")
    dominant, other = (r_fused, r_bwd) if t_fused >= t_bwd else (r_bwd, r_fused)
    dominant["other_pass"] = other
    if t_fused > 0 and t_bwd > 0:
        dominant["score_fwd_bwd_algorithmic_tflops"] = 6.0 * b2d / (t_fused + t_bwd) / 1e12
        dominant["score_fwd_bwd_frac"] = dominant["score_fwd_bwd_algorithmic_tflops"] / MFMA_F32_PEAK_TFLOPS
    # gather + scatter (HBM): algorithmic bytes per step: gather 16BD+16B, SGD 24BD, Adagrad 40BD
    gs_bytes = 16 * batch * dim + 16 * batch + (24 if args.optimizer == "sgd" else 40) * batch * dim
    if cfg.n_category_buckets:         # +8BD per extra hashed feature: its rows summed into the item input
        gs_bytes += 8 * batch * dim

    def per_step(tag):                 # ms per STEP (a step may launch a tagged kernel more than once)
        return sum(prof[tag]) / steps_of[tag]
    # K2 apply: in the step it is half of the single optimizer launch; its own duration comes from the un-fused detail pass
    apply_ms = (sum(unfused["sparse_apply"]) / detail_steps) if unfused.get("sparse_apply") else per_step("sparse_apply")
    t_gs_noplan = max((per_step("gather") + apply_ms) * 1e-3 + max(lookup_us or 0.0, 0.0) * 1e-6, 1e-12)
    plan_on_main = not trainer.plan_on_side_stream
    plan_bytes = 20 * batch * (2 + (1 if cfg.n_category_buckets else 0))     # id read + sorted id + position written, per table
    # the plan is on the step's critical path when it runs on the main stream (the default): then it is COUNTED
    fused_sort = trainer.one_launch_optimizer(batch)      # (after the timed steps: what the skew probe left the trainer on)
    plan_ms = (sum(unfused["sparse_plan"]) / detail_steps) if unfused.get("sparse_plan") else per_step("sparse_plan")
    if fused_sort:
        # no plan launch: the optimizer launch sorts each row range in LDS and applies the update itself.  Its WHOLE duration
        # is counted (the dense tower update runs beside the sorting workgroups inside it and ends first)
        plan_bytes = 8 * batch * (2 + (1 if cfg.n_category_buckets else 0))  # the ids, read once per table (algorithmic)
        t_gs = max(per_step("optimizer") * 1e-3 + max(lookup_us or 0.0, 0.0) * 1e-6, 1e-12)
        gs_bytes += plan_bytes
    else:
        t_gs = t_gs_noplan + (plan_ms * 1e-3 if plan_on_main else 0.0)
        if plan_on_main:
            gs_bytes += plan_bytes
    rp_opt = rocprof_avg_us("optimizer_ids_kernel") if (args.config == "cfg3" and args.optimizer == "sgd" and fused_sort) else None
    rp_lookup = rocprof_lookup_share_us() if (args.config == "cfg3" and args.optimizer == "sgd" and fused_sort and lookup_fused) else None
    out = {
        "metric": "user-item pairs/sec (train step) + embedding-gather HBM GB/s, 1/2/4/8 MI355X",
        "value": batch / (dt / args.steps), "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {n_users} users x {n_items} items, emb_dim {dim}, towers "
                               f"{dim}->{'->'.join(map(str, tower_dims))}, batch {batch}, in-batch sampled softmax T=0.1, "
                               f"{args.optimizer} lr 1e-3, ids {args.ids}"
                               + (f", + {cfg.n_category_buckets}-bucket hashed category feature summed into the item input"
                                  if cfg.n_category_buckets else ""),
                   "global_batch": batch, "parallelism": "single GPU"},
        "timing_note": f"kernel durations = the dispatch's own begin/end timestamps (hipExtLaunchKernelGGL event pair: what "
                       f"rocprofv3 --kernel-trace reports; no barrier packets). Inside the timed region: {timed_tags} only, every "
                       f"{stride}th step ({len(prof['score_fused'])} samples: a timestamped dispatch still costs the stream ~4 us); "
                       f"every other kernel from an untimed detail pass of {detail_steps} further steps",
        "roofline": dominant,
        "roofline_hbm": {"bound": "hbm",
                         "kernel": "K1 + K2 on the critical path: "
                                   + ("the embedding lookup fused into the first tower layer's GEMM "
                                      "loaders (time = what the layer-0 fwd and bwd launches cost MORE than on a materialised "
                                      "input, from an un-fused detail pass) + " if lookup_fused else
                                      "gather2 (the lookup as its own launch, both towers: gather_us) + ")
                                   + ""
                                   + ("optimizer_ids_kernel, the step's ONE optimizer launch, whole duration: every table's ids "
                                      "are sorted per row range in LDS by the workgroups that then sum the duplicate gradient rows "
                                      "and apply fused SGD/Adagrad to exactly those rows (no plan launch, no sorted ids in HBM); "
                                      "the dense tower update runs in the same launch beside them. " if fused_sort else
                                      "the sparse apply (segmented sums, fused SGD/Adagrad, arrival-ticket finish; all tables; timed "
                                      "as its own launch in the un-fused detail pass - in the step it shares ONE launch with the "
                                      "dense tower update: optimizer_launch_us) + "
                                      + ("part_sort_kernel (the sort plan: one launch for all tables, on the main stream in front "
                                         "of the forward pass). " if plan_on_main else
                                         "EXCLUDED and reported beside it: part_sort_kernel (the plan: one launch for all tables on "
                                         "a side stream, concurrent with the forward pass; TT_PLAN_STREAM=side). "))
                                   + ("Row-range id lists (r04): the forward lookup appends every id to the list of the row range "
                                      "its sorting workgroup owns; the optimizer launch reads its ~64 entries instead of scanning all ids. "
                                      if (fused_sort and lookup_fused and os.environ.get("TT_ID_BUCKETS", "1") != "0") else "")
                                   + "Durations are dispatch timestamps (= rocprofv3's); an EMPTY kernel takes 4.0 us by the same "
                                     "clock on this system (profiles/r04_launch_floor.txt)",
                         "achieved": gs_bytes / t_gs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gs_bytes / t_gs / 1e9 / HBM_PEAK_GBS,
                         "traffic": None,
                         "lookup_in_gemm_us": lookup_us,
                         "unfused_gather2_us": (mean(unfused["gather"]) * 1e3) if unfused.get("gather") else None,
                         "gather_us": per_step("gather") * 1e3,
                         "sparse_apply_us": apply_ms * 1e3,
                         "optimizer_launch_us": per_step("optimizer") * 1e3,
                         "rocprof_optimizer_launch_us": rp_opt,
                         "frac_with_rocprof_launch": (gs_bytes / ((rp_opt + max(lookup_us or 0.0, 0.0)) * 1e-6) / 1e9 / HBM_PEAK_GBS) if rp_opt else None,
                         "rocprof_source": f"{KSTATS_FILE} (committed rocprofv3 duration of the optimizer launch; not re-measured in this run)",
                         "lookup_share_rocprof_ab_us": rp_lookup,
                         "frac_with_rocprof_ab_share": (gs_bytes / ((per_step("optimizer") * 1e3 + rp_lookup) * 1e-6) / 1e9 / HBM_PEAK_GBS) if rp_lookup else None,
                         "lookup_share_source": "profiles/r04_lookup_share.json (committed rocprofv3 A/B of the layer-0 launches, lookup fused vs "
                                                "materialised input, same box, alternating; not re-measured in this run) - frac_with_rocprof_ab_share = "
                                                "this run's optimizer launch + that share",
                         "unfused_dense_update_us": (mean(unfused["dense_update"]) * 1e3) if unfused.get("dense_update") else None,
                         "sparse_plan_us": plan_ms * 1e3,
                         "sparse_plan_stream": "inside the optimizer launch (un-fused detail pass figure above)" if fused_sort
                                               else ("main" if plan_on_main else "side"),
                         "sort_fused_into_optimizer_launch": bool(fused_sort),
                         "skew_probe": {"largest_row_range_load": int(trainer.range_load), "limit": int(trainer.skew_limit),
                                        "path": "one launch" if fused_sort else "plan + optimizer step"},
                         "frac_without_plan": (gs_bytes - (plan_bytes if (plan_on_main or fused_sort) else 0)) / t_gs_noplan / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes": gs_bytes},
        "loss_per_pair": loss / batch,
    }
    # the tower GEMMs (K3): forward + dx + dw of every layer of both towers = 3 * 2*B*sum(in*out) * 2 FLOPs,
    # over the timestamped launches of the untimed detail pass
    dims_all = [dim] + list(cfg.user_dims)
    gemm_flops = 3.0 * 2.0 * batch * sum(a * b for a, b in zip(dims_all[:-1], dims_all[1:])) * 2.0
    t_gemm = (per_step("dense_fwd") + per_step("dense_bwd")) * 1e-3
    if t_gemm > 0 and cfg.symmetric:
        out["roofline_gemm"] = {
            "bound": "mfma", "kernel": "tower GEMMs: " + ("tower_fwd2_kernel (both layers of both towers, one launch) + " if
                                                           len(tower_dims) == 2 and ops.tower_fwd2_supported(batch, *dims_all) else
                                                           "gemm_kernel per layer (both towers per launch) + ")
                                       + "gemm_bwd_kernel per layer (dx + dw + db tiles of both towers in one launch)",
            "achieved": gemm_flops / t_gemm / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": gemm_flops / t_gemm / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": None,
            "launches_per_step": (len(prof["dense_fwd"]) + len(prof["dense_bwd"])) // max(detail_steps, 1),
            "us_per_step": t_gemm * 1e6, "algorithmic_gflop": gemm_flops / 1e9, "dtype": "f32-input MFMA"}
        if args.config == "cfg3" and args.optimizer == "sgd":
            rp = rocprof_avg_us("tower_fwd2_kernel", "gemm_bwd_kernel<0>", "gemm_bwd_kernel<256>")
            if rp:
                out["roofline_gemm"].update({"rocprof_us_per_step": rp, "rocprof_frac": gemm_flops / (rp * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                             "rocprof_source": f"{KSTATS_FILE} (committed; not re-measured in this run - a cross-check of the live "
                                                               "dispatch-timestamp figure)"})
    if args.config == "ref":
        out["note"] = out_note_ref
        out["config"]["dropout_rate"] = REF_DROPOUT
    if alt is not None:
        alt_steps, dta, tf_, tb_, aloss = alt
        # per launch of the FUSED pass: GEMM1 = 2*B^2*D algorithmic FLOPs at 6 bf16 products each, GEMM2 = 2*B^2*D at 3
        peak_equiv = 2500.0 * 4.0 / 18.0
        a_alg = 4.0 * b2d / tf_ / 1e12
        out["roofline_alt"] = {
            "bound": "mfma", "dtype": "bf16x3 (f32-emulated: x = hi + mid + lo bf16 pieces, f32 accumulate, f32 softmax)",
            "kernel": f"score_kernel<{sd},FUSED,bf16x3> (loss + dq pass on v_mfma_f32_32x32x16_bf16; algorithmic 4*B^2*D, "
                      "executed 18*B^2*D bf16 FLOPs: 6 products per logit term, 3 per gradient term)",
            "achieved": a_alg, "peak": peak_equiv, "unit": "TFLOP/s (f32-equivalent)", "frac": a_alg / peak_equiv,
            "peak_note": "2500 TFLOP/s dense bf16 MFMA x 4/18 (algorithmic / executed products)",
            "executed_bf16_tflops": 18.0 * b2d / tf_ / 1e12, "executed_frac_of_bf16_peak": 18.0 * b2d / tf_ / 1e12 / 2500.0,
            "avg_launch_us": tf_ * 1e6, "other_pass_avg_launch_us": tb_ * 1e6, "traffic": None,
            "value_alt": batch / (dta / alt_steps), "unit_alt": "pairs/s", "ms_per_step_alt": dta / alt_steps * 1e3,
            "steps_alt": alt_steps, "loss_per_pair_alt": aloss / batch,
            "note": "whole train step with scorer_precision='bf16x3'; NOT the headline (value / roofline are exact f32)"}
    if not args.no_cpu_baseline and not cfg.n_category_buckets:     # the torch-CPU port covers the cfg1-cfg4 model
        out["cpu_baseline"] = cpu_baseline(trainer, cfg, seed, args, batch)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
