"""Time the backward tower launches of cfg3 as separate pieces (hipEvents over back-to-back calls): dx only / dw only per layer
vs the fused dx+dw launches - the premise check for a chained-dx kernel + one dW launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_amazon_recommender_amd import ops
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer

dev = torch.device("cuda:0")
cfg = TwoTowerConfig(n_users=5_000_000, n_items=10_000_000, embedding_dim=128, tower_dims=[256, 128], batch_size=8192)
tr = TwoTowerTrainer(cfg, dev, seed=3)
u, i = tr.synthetic_batch(3, 0)
for _ in range(3):
    tr.step(u, i)
ut, it = tr.user_tower, tr.item_tower
none2 = (None, None)
lks = tr._lookups(u, i, None)


def timed(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


bits1 = (ut.bits[1], it.bits[1])
pieces = {
    "L1 dx+dw": lambda: ops.dense_bwd2((ut.acts[1], it.acts[1]), (ut.w[1], it.w[1]), (ut.dz[1], it.dz[1]), (ut.dz[0], it.dz[0]), none2,
                                       (ut.dw_slabs[1], it.dw_slabs[1]), (ut.db_slabs[1], it.db_slabs[1]), dx_relu_bits=bits1),
    "L1 dx only": lambda: ops.dense_bwd2((ut.acts[1], it.acts[1]), (ut.w[1], it.w[1]), (ut.dz[1], it.dz[1]), (ut.dz[0], it.dz[0]), none2,
                                         none2, none2, dx_relu_bits=bits1),
    "L1 dw only": lambda: ops.dense_bwd2((ut.acts[1], it.acts[1]), (ut.w[1], it.w[1]), (ut.dz[1], it.dz[1]), none2, none2,
                                         (ut.dw_slabs[1], it.dw_slabs[1]), (ut.db_slabs[1], it.db_slabs[1])),
    "L0 dx+dw (lookup)": lambda: ops.dense_bwd2((None, None), (ut.w[0], it.w[0]), (ut.dz[0], it.dz[0]), (ut.demb, it.demb), none2,
                                                (ut.dw_slabs[0], it.dw_slabs[0]), (ut.db_slabs[0], it.db_slabs[0]), lookups=lks),
    "L0 dx only": lambda: ops.dense_bwd2((ut.demb, it.demb), (ut.w[0], it.w[0]), (ut.dz[0], it.dz[0]), (ut.demb, it.demb), none2, none2, none2),
    "L0 dw only (lookup)": lambda: ops.dense_bwd2((None, None), (ut.w[0], it.w[0]), (ut.dz[0], it.dz[0]), none2, none2,
                                                  (ut.dw_slabs[0], it.dw_slabs[0]), (ut.db_slabs[0], it.db_slabs[0]), lookups=lks),
    "fwd fused (for scale)": lambda: tr.towers_forward(u, i) if hasattr(tr, "towers_forward") else None,
}
for name, fn in pieces.items():
    try:
        print(f"{name:24s} {timed(fn):7.2f} us (back-to-back launches, incl. ~1-2 us launch gap)")
    except Exception as e:  # noqa: BLE001
        print(name, "failed:", repr(e)[:200])
