"""debug: optimizer_step_ids vs plan + step on the shapes of test_train_cli_distributed_one_rank_equals_single_gpu"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth
from two_tower_amazon_recommender_amd import ops
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for (n, dim, rows3, npad, opt) in [(512, 32, [2000, 1500, 30], 208, "sgd"), (512, 32, [2000, 1500, 30], 0, "sgd"), (512, 32, [2000, 1500, 30], 208, "adagrad"),
                                   (1024, 64, [3000, 100, 30], 300, "sgd"), (8192, 128, [100000, 5000, 30], 1000, "sgd")]:
    rng = np.random.default_rng(11)
    ids = []
    for r in rows3:
        x = rng.integers(0, r, n).astype(np.int64)
        if npad: x[n - npad:] = -1
        ids.append(T(x))
    grads = [T(synth.uniform_f32(6, 9 + t, n * dim, -1.0, 2.0).reshape(n, dim)) for t in range(3)]
    grads[2] = grads[1]
    wslab = T(synth.uniform_f32(6, 20, 4 * 1000, -1.0, 2.0).reshape(4, 1000))
    def state():
        tabs = [T(synth.embedding_table(7, 1 + t, r, dim)) for t, r in enumerate(rows3)]
        accs = [torch.full_like(x, 0.1) if opt == "adagrad" else None for x in tabs]
        w = T(synth.uniform_f32(7, 30, 1000, -1.0, 2.0)); wacc = torch.full_like(w, 0.1) if opt == "adagrad" else None
        return tabs, accs, w, wacc
    plans = [ops.SparsePlan(n, dev) for _ in range(3)]
    ta, aa, wa, wacca = state()
    ops.sparse_plan_batched(plans, ids, rows3)
    ops.optimizer_step_(opt, [(ta[t], aa[t], grads[t], plans[t]) for t in range(3)], [ops.make_dense_seg(wa, wacca, wslab, 4, 1e-6)], 0.01, 1e-7)
    tb, ab, wb, waccb = state()
    ops.optimizer_step_ids_(opt, [(tb[t], ab[t], grads[t], ids[t], plans[t]) for t in range(3)], [ops.make_dense_seg(wb, waccb, wslab, 4, 1e-6)], 0.01, 1e-7)
    for t in range(3):
        bad = (ta[t] != tb[t]).any(1).nonzero().flatten().cpu().numpy()
        x = ids[t].cpu().numpy()
        cnt = np.bincount(x[x >= 0], minlength=rows3[t])
        print(f"n {n} dim {dim} pad {npad} {opt} table {t} rows {rows3[t]}: {len(bad)} rows differ; multiplicities of differing rows: {np.bincount(cnt[bad])[:12].tolist() if len(bad) else []}; of all touched rows: {np.bincount(cnt)[:12].tolist()}")
        if len(bad):
            r0 = bad[0]; print("   first bad row", r0, "mult", cnt[r0], "plan+step", ta[t][r0, :4].tolist(), "ids", tb[t][r0, :4].tolist())
