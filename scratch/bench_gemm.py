import sys, time, torch
sys.path.insert(0, "/root/repo")
from two_tower_amazon_recommender_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (m, k, n) in [(8192, 128, 256), (8192, 256, 128)]:
    x = torch.randn(m, k, device=dev); w = torch.randn(k, n, device=dev); b = torch.randn(n, device=dev)
    y = torch.empty(m, n, device=dev); dz = torch.randn(m, n, device=dev); dx = torch.empty(m, k, device=dev)
    ns = ops.dense_bwd_num_slabs(m); dw = torch.empty(ns, k, n, device=dev); db = torch.empty(ns, n, device=dev)
    print(f"m{m} k{k} n{n}: fwd {t(lambda: ops.dense_fwd(x, w, b, True, out=y)):.1f} us  bwd(dx+dw) {t(lambda: ops.dense_bwd(x, w, dz, dx, x, dw, db)):.1f} us  torch.mm {t(lambda: torch.mm(x, w, out=y)):.1f} us")
