"""Sort-plan microbench: plan of 2 tables x n ids, uniform (U) and Zipf (Z) ids; TT_SORT_GROUPS picks the partition count."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_amazon_recommender_amd import ops  # noqa: E402
from bench_k2 import timed  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for n, rows_u, rows_i in ((8192, 100_000, 100_000), (8192, 5_000_000, 10_000_000), (16384, 5_000_000, 100_000_000), (4096, 1_000_000, 1_000_000)):
        for variant in ("U", "Z"):
            iu = torch.empty(n, dtype=torch.int64, device=dev); ii = torch.empty(n, dtype=torch.int64, device=dev)
            ops.fill_ids_(iu, 1, 3, rows_u, variant); ops.fill_ids_(ii, 1, 4, rows_i, variant)
            pu, pi = ops.SparsePlan(n, dev), ops.SparsePlan(n, dev)
            t2 = timed(lambda: ops.sparse_plan_batched([pu, pi], [iu, ii], [rows_u, rows_i]), 200)
            t1 = timed(lambda: pu.run(iu, rows_u), 200)
            print(json.dumps({"groups": os.environ.get("TT_SORT_GROUPS", "auto"), "n": n, "rows": [rows_u, rows_i], "ids": variant,
                              "plan_2tables_us": round(t2, 2), "plan_1table_us": round(t1, 2)}), flush=True)


if __name__ == "__main__":
    main()
