"""Round 4: the owner-side update of the row-sharded step at N = 8 (weak cfg3: 8 ranks x 2 tables x 2048 capacity = 32768 received slots,
about half of them -1 padding, one combined shard of (5M + 10M) x 8 / 8 rows, dim 128) - the long-list one-launch kernel against
plan + step, kernel durations from rocprofv3 (run under scratch/r04_owner_shape_prof.sh)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd import ops
dev = torch.device("cuda:0")
rows, dim, n = 15_000_000, 128, 32768
rng = np.random.default_rng(2)
for opt in ("sgd", "adagrad"):
    for fill in (0.5, 0.25, 1.0):
        ids = rng.integers(0, rows, n).astype(np.int64)
        ids[rng.random(n) > fill] = -1
        ids_d = torch.from_numpy(ids).to(dev)
        grads = torch.randn(n, dim, device=dev)
        table = torch.randn(rows, dim, device=dev)
        acc = torch.full_like(table, 0.1) if opt == "adagrad" else None
        w = torch.randn(132_000, device=dev); wacc = torch.full_like(w, 0.1) if opt == "adagrad" else None
        slab = torch.randn(1, 132_000, device=dev)
        seg = [ops.make_dense_seg(w, wacc, slab, 1, 1e-6)]
        plan = ops.SparsePlan(n, dev)
        for _ in range(12):
            ops.sparse_plan_batched([plan], [ids_d], [rows])
            ops.optimizer_step_(opt, [(table, acc, grads, plan)], seg, 0.01, 1e-7)
        torch.cuda.synchronize()
        for _ in range(12):
            ops.optimizer_step_ids_(opt, [(table, acc, grads, ids_d, plan)], seg, 0.01, 1e-7)
        torch.cuda.synchronize()
        del table, acc
        torch.cuda.empty_cache()
