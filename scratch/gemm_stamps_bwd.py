"""Per-workgroup phase stamps of the two tower backward launches of the cfg3 step, in the hot path's form (ReLU sign bits
as the dx mask, fused lookup in layer 0's dW): debug build with -DTT_GEMM_STAMPS through TT_LIB_PATH.
Stamps: 0 start, 1 first k-tile in LDS (first barrier passed), 2 MFMA loop done, 3 stores drained."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_amazon_recommender_amd import _lib, ops
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer

dev = torch.device("cuda:0")
cfg = TwoTowerConfig(n_users=5_000_000, n_items=10_000_000, embedding_dim=128, tower_dims=[256, 128], batch_size=8192)
tr = TwoTowerTrainer(cfg, dev, seed=3)
u, i = tr.synthetic_batch(3, 0)
for _ in range(3):
    tr.step(u, i)
ut, it = tr.user_tower, tr.item_tower
lib = _lib.load()
lib.tt_debug_gemm_stamps.restype = C.c_int
lib.tt_debug_gemm_stamps.argtypes = [C.c_void_p, C.c_int]
none2 = (None, None)


def run(name, fn, nwg):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 4, dtype=np.uint64)
    assert lib.tt_debug_gemm_stamps(buf.ctypes.data, buf.size) == 0
    s = buf.reshape(8192, 4)[:nwg].astype(np.int64)
    s = s[s[:, 3] > 0]
    t0 = s[:, 0].min()
    us = (s - t0) / 100.0
    ph = np.stack([us[:, 0], us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]], 1)
    pc = lambda a: " ".join(f"{np.percentile(a, q):.1f}" for q in (0, 10, 50, 90, 100))
    print(f"{name}: {len(s)} WGs; start p0/10/50/90/100 {pc(us[:, 0])} | load->LDS {pc(ph[:, 1])} | MFMA loop {pc(ph[:, 2])} | store drain {pc(ph[:, 3])} | end {pc(us[:, 3])}")
    # by kind (r04): workgroups whose MFMA loop is in the longer / shorter half = the two kinds of tile
    med = (ph[:, 2].max() + ph[:, 2].min()) / 2
    for kind, sel in (("long-k tiles", ph[:, 2] >= med), ("short-k tiles", ph[:, 2] < med)):
        if sel.sum():
            print(f"   {kind}: {int(sel.sum())} WGs; start {pc(us[sel, 0])} | load->LDS {pc(ph[sel, 1])} | MFMA loop {pc(ph[sel, 2])} | store drain {pc(ph[sel, 3])} | end {pc(us[sel, 3])}")
    # by kind: the launch puts one kind of tile first; split the workgroups at the largest jump of MFMA-loop length
    order = np.argsort(ph[:, 2])
    print("   MFMA-loop length histogram (us):", np.histogram(ph[:, 2], bins=8)[0].tolist(), np.histogram(ph[:, 2], bins=8)[1].round(1).tolist())


lks = tr._lookups(u, i, None)
run("bwd L1 dx+dw", lambda: ops.dense_bwd2((ut.acts[1], it.acts[1]), (ut.w[1], it.w[1]), (ut.dz[1], it.dz[1]), (ut.dz[0], it.dz[0]),
                                           none2, (ut.dw_slabs[1], it.dw_slabs[1]), (ut.db_slabs[1], it.db_slabs[1]),
                                           dx_relu_bits=(ut.bits[1], it.bits[1])), 4096)
run("bwd L0 dx+dw (lookup)", lambda: ops.dense_bwd2((None, None), (ut.w[0], it.w[0]), (ut.dz[0], it.dz[0]), (ut.demb, it.demb),
                                                    none2, (ut.dw_slabs[0], it.dw_slabs[0]), (ut.db_slabs[0], it.db_slabs[0]),
                                                    lookups=lks), 4096)
