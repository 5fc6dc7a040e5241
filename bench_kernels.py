"""Kernel microbenchmarks at sizes where launch latency does not dominate (SURVEY.md §7 "provide a large-B
microbench"): embedding gather and fused sparse SGD / Adagrad over 1M ids on a 10M x 128 table, and the fused
scorer at several batch sizes.  Prints one JSON line per kernel: algorithmic bytes (or FLOPs), time from hipEvents
over back-to-back launches, and the fraction of the MI355X roofline (HBM 8.0 TB/s spec, f32 MFMA 157.3 TF).

    python bench_kernels.py [--rows 10000000] [--ids 1048576] [--dim 128]
"""
import argparse
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from two_tower_amazon_recommender_amd import ops  # noqa: E402

HBM, MFMA = 8000.0, 157.3


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--ids", type=int, default=1 << 20)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", choices=("all", "table", "score"), default="all", help="table: gather / sparse kernels; score: scorer")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows, n, d = args.rows, args.ids, args.dim
    table = torch.empty(rows, d, device=dev)
    ops.fill_uniform_(table, 1, 1, -0.05, 0.1)
    accum = torch.full_like(table, 0.1)
    grads = torch.empty(n, d, device=dev)
    ops.fill_uniform_(grads, 1, 9, -1.0, 2.0)
    out = torch.empty(n, d, device=dev)
    for variant in (("U", "Z") if args.only != "score" else ()):
        ids = torch.empty(n, dtype=torch.int64, device=dev)
        ops.fill_ids_(ids, 1, 3, rows, variant)
        uniq = int(torch.unique(ids).numel())
        t = timed(lambda: ops.embedding_gather(table, ids, out=out), args.iters)
        gb = (8 * d * n + 8 * n) / 1e9
        print(json.dumps({"kernel": "gather_kernel", "ids": variant, "n_ids": n, "rows": rows, "dim": d, "us": t * 1e6,
                          "algorithmic_GB": gb, "GBps": gb / t, "frac_hbm_8TBs": gb / t / HBM}))
        plan = ops.SparsePlan(n, dev).run(ids, rows)
        t = timed(lambda: ops.sparse_sgd_(table, grads, plan, 1e-6), args.iters)
        gb = (4 * d * n + 8 * d * uniq + 12 * n) / 1e9          # every grad row read once; distinct rows read+written
        print(json.dumps({"kernel": "sparse_apply_kernel<SGD>", "ids": variant, "n_ids": n, "distinct": uniq, "us": t * 1e6,
                          "algorithmic_GB": gb, "GBps": gb / t, "frac_hbm_8TBs": gb / t / HBM}))
        t = timed(lambda: ops.sparse_adagrad_(table, accum, grads, plan, 1e-6), args.iters)
        gb = (4 * d * n + 16 * d * uniq + 12 * n) / 1e9
        print(json.dumps({"kernel": "sparse_apply_kernel<Adagrad>", "ids": variant, "n_ids": n, "distinct": uniq, "us": t * 1e6,
                          "algorithmic_GB": gb, "GBps": gb / t, "frac_hbm_8TBs": gb / t / HBM}))
        t = timed(lambda: plan.run(ids, rows), args.iters)
        print(json.dumps({"kernel": "tt_sparse_plan (LDS chunk sort + merge <= 262144 ids, rocPRIM radix sort beyond)", "ids": variant,
                          "n_ids": n, "us": t * 1e6}))
    del table, accum, grads, out
    for b, dd in (((4096, 64), (8192, 128), (16384, 128), (8192, 256)) if args.only != "table" else ()):
        q = torch.empty(b, dd, device=dev); c = torch.empty(b, dd, device=dev)
        ops.fill_uniform_(q, 2, 1, -0.3, 0.6); ops.fill_uniform_(c, 2, 2, -0.3, 0.6)
        ws = torch.empty(ops.retrieval_workspace_bytes(b, b, dd), dtype=torch.uint8, device=dev)
        lse = torch.empty(b, device=dev); pr = torch.empty(b, device=dev); loss = torch.empty(1, device=dev)
        dq = torch.empty(b, dd, device=dev); dc = torch.empty(b, dd, device=dev)
        t = timed(lambda: ops.retrieval_fwd_bwd(q, c, 10.0, ws, lse, pr, loss, dq, dc), 10)
        tf = 6.0 * b * b * dd / 1e12
        print(json.dumps({"kernel": "tt_retrieval_fwd_bwd_f32 (2 score passes + combines)", "batch": b, "dim": dd, "us": t * 1e6,
                          "algorithmic_TFLOP": tf, "algorithmic_TFLOPs": tf / t, "frac_mfma_f32": tf / t / MFMA,
                          "executed_TFLOPs": 6.0 * b * b * dd / 1e12 / t,
                          "note": "executed = algorithmic since r02: the dc pass reads the stored dot products back (no second GEMM1)"}))


if __name__ == "__main__":
    main()
