"""Round 4: the long-list optimizer launch (16,385 .. 65,536 ids per table) against plan + optimizer step, per shape (us, median of 20)."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd import ops
from oracle import synth   # (ids only: the generator of the test inputs)

dev = torch.device("cuda:0")
def T(x): return torch.from_numpy(np.ascontiguousarray(x)).to(dev)

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))

cases = [(32768, 256, (4_000_000, 3_000_000), "Z", "adagrad"), (32768, 256, (4_000_000, 3_000_000), "U", "adagrad"),
         (32768, 128, (10_000_000, 5_000_000), "U", "sgd"), (65536, 128, (10_000_000, 5_000_000), "Z", "sgd"),
         (20000, 128, (5_000_000, 100_000), "U", "sgd"), (32768, 128, (1000, 1000), "U", "sgd"),
         (40000, 64, (5, 1), "U", "sgd"), (65536, 128, (3, 1), "U", "adagrad")]
import os as _os
_sel = _os.environ.get('TT_BIG_CASES')
if _sel: cases = [cases[int(i)] for i in _sel.split(',')]
for n, dim, rows, kind, opt in cases:
    rng = np.random.default_rng(1)
    ids = [T((synth.ids_powerlaw(5, 3 + t, n, r) if kind == "Z" else rng.integers(0, r, n)).astype(np.int64)) for t, r in enumerate(rows)]
    grads = [torch.randn(n, dim, device=dev) for _ in rows]
    tabs = [torch.randn(r, dim, device=dev) for r in rows]
    accs = [torch.full_like(x, 0.1) if opt == "adagrad" else None for x in tabs]
    w = torch.randn(100_000, device=dev); wacc = torch.full_like(w, 0.1) if opt == "adagrad" else None
    wslab = torch.randn(4, 100_000, device=dev)
    plans = [ops.SparsePlan(n, dev) for _ in rows]
    seg = [ops.make_dense_seg(w, wacc, wslab, 4, 1e-6)]
    def plan_step():
        ops.sparse_plan_batched(plans, ids, list(rows))
        ops.optimizer_step_(opt, [(tabs[t], accs[t], grads[t], plans[t]) for t in range(2)], seg, 0.01, 1e-7)
    def one():
        ops.optimizer_step_ids_(opt, [(tabs[t], accs[t], grads[t], ids[t], plans[t]) for t in range(2)], seg, 0.01, 1e-7)
    import os
    def plan_only():
        ops.sparse_plan_batched(plans, ids, list(rows))
    pn = timed(plan_only)
    os.environ["TT_SORT_CHUNKS"] = "1"
    pc = timed(plan_only); a_old = timed(plan_step)
    del os.environ["TT_SORT_CHUNKS"]
    a = timed(plan_step); b = timed(one)
    uniq = sum(int(torch.unique(i).numel()) for i in ids)
    mb = (uniq * dim * 4 * (2 if opt == "sgd" else 4) + 2 * n * dim * 4) / 1e6
    print(json.dumps({"n_ids": n, "dim": dim, "rows": rows, "ids": kind, "opt": opt, "plan_us": round(pn, 1), "plan_r03_chunks_us": round(pc, 1), "plan_then_step_r03_us": round(a_old, 1), "plan_then_step_us": round(a, 1), "one_launch_us": round(b, 1),
                      "algorithmic_MB": round(mb, 1), "one_launch_frac_of_8TBs": round(mb / 8e6 / (b * 1e-6) , 3)}), flush=True)
