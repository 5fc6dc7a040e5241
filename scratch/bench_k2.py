"""Microbench of the K1/K2 kernels at the TRAINING batch (cfg3: 8192 ids per table): sort plan (all tables in one
launch), fused sparse apply, gather.  hipEvent time over back-to-back launches; prints JSON lines."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from two_tower_amazon_recommender_amd import ops  # noqa: E402


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    d = 128
    for n, rows_u, rows_i in ((8192, 5_000_000, 10_000_000), (16384, 5_000_000, 100_000_000), (4096, 1_000_000, 1_000_000)):
        tu = torch.empty(rows_u, d, device=dev); ti = torch.empty(rows_i, d, device=dev)
        ops.fill_uniform_(tu, 1, 1, -0.05, 0.1); ops.fill_uniform_(ti, 1, 2, -0.05, 0.1)
        g = torch.empty(2, n, d, device=dev); ops.fill_uniform_(g, 1, 9, -1.0, 2.0)
        for variant in ("U", "Z"):
            iu = torch.empty(n, dtype=torch.int64, device=dev); ii = torch.empty(n, dtype=torch.int64, device=dev)
            ops.fill_ids_(iu, 1, 3, rows_u, variant); ops.fill_ids_(ii, 1, 4, rows_i, variant)
            pu, pi = ops.SparsePlan(n, dev), ops.SparsePlan(n, dev)
            t_plan2 = timed(lambda: ops.sparse_plan_batched([pu, pi], [iu, ii], [rows_u, rows_i]))
            t_plan1 = timed(lambda: pu.run(iu, rows_u))
            t_apply = timed(lambda: ops.sparse_update2_("sgd", tu, None, g[0], pu, ti, None, g[1], pi, 1e-6))
            ou, oi = torch.empty(n, d, device=dev), torch.empty(n, d, device=dev)
            t_gather = timed(lambda: ops.embedding_gather2(tu, iu, ou, ti, ii, oi))
            print(json.dumps({"n": n, "ids": variant, "plan_2tables_us": t_plan2, "plan_1table_us": t_plan1,
                              "apply2_sgd_us": t_apply, "gather2_us": t_gather}), flush=True)
        del tu, ti


if __name__ == "__main__":
    main()
