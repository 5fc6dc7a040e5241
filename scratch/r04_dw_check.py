"""dW / db slabs of tt_dense_bwd (and the fused-lookup form) against a float64 torch product, per slab."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_amazon_recommender_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (m, k, n) in ((8192, 128, 256), (8192, 256, 128), (4096, 64, 64), (1024, 128, 512)):
    x = torch.randn(m, k, device=dev); w = torch.randn(k, n, device=dev) * 0.1; dz = torch.randn(m, n, device=dev)
    ns = ops.dense_bwd_num_slabs(m)
    dw = torch.empty(ns, k, n, device=dev); db = torch.empty(ns, n, device=dev); dx = torch.empty(m, k, device=dev)
    ops.dense_bwd(x, w, dz, dx, None, dw, db)
    rows = (m + ns - 1) // ns
    rows = (rows + 31) // 32 * 32
    worst = 0.0; worst_b = 0.0
    for s in range(ns):
        xs, ds = x[s * rows:(s + 1) * rows].double(), dz[s * rows:(s + 1) * rows].double()
        ref = xs.t() @ ds
        worst = max(worst, ((dw[s].double() - ref).abs().max() / ref.abs().max()).item())
        rb = ds.sum(0)
        worst_b = max(worst_b, ((db[s].double() - rb).abs().max() / rb.abs().max()).item())
    dxr = dz.double() @ w.double().t()
    print(f"m {m} k {k} n {n} slabs {ns}: dW max rel err {worst:.3e}  db {worst_b:.3e}  dx {((dx.double()-dxr).abs().max()/dxr.abs().max()).item():.3e}")
    # fused lookup form: x = table[ids]
    table = torch.randn(50000, k, device=dev); ids = torch.randint(0, 50000, (m,), device=dev)
    lk = ops.make_lookup(table, ids)
    dw2 = torch.empty_like(dw); db2 = torch.empty_like(db)
    ops.dense_bwd(None, w, dz, dx, None, dw2, db2, lookup=lk)
    xg = table[ids]
    worst = 0.0
    for s in range(ns):
        ref = xg[s * rows:(s + 1) * rows].double().t() @ dz[s * rows:(s + 1) * rows].double()
        worst = max(worst, ((dw2[s].double() - ref).abs().max() / ref.abs().max()).item())
    print(f"   lookup form: dW max rel err {worst:.3e}")
