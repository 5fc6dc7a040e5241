"""Per-workgroup phase timestamps of the tower GEMM launches (debug build with -DTT_GEMM_STAMPS, TT_LIB_PATH):
s_memrealtime (100 MHz) at: 0 start, 1 first k-tile in LDS (first barrier passed), 2 MFMA loop done, 3 stores drained."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from two_tower_amazon_recommender_amd import _lib, ops
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer

dev = torch.device("cuda:0")
cfg = TwoTowerConfig(n_users=100_000, n_items=100_000, embedding_dim=128, tower_dims=[256, 128], batch_size=8192)
tr = TwoTowerTrainer(cfg, dev, seed=3)
u, i = tr.synthetic_batch(3, 0)
for _ in range(3):
    tr.step(u, i)
ut, it = tr.user_tower, tr.item_tower
lib = _lib.load()
lib.tt_debug_gemm_stamps.restype = C.c_int
lib.tt_debug_gemm_stamps.argtypes = [C.c_void_p, C.c_int]


def run(name, fn, nwg):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 4, dtype=np.uint64)
    assert lib.tt_debug_gemm_stamps(buf.ctypes.data, buf.size) == 0
    s = buf.reshape(8192, 4)[:nwg].astype(np.int64)
    t0 = s[:, 0].min()
    us = (s - t0) / 100.0
    ph = np.stack([us[:, 0], us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]], 1)
    print(f"{name}: {nwg} WGs; start spread {us[:,0].max():.2f} us; per-WG mean [start, load->LDS, MFMA loop, store drain] = "
          f"{ph.mean(0).round(2).tolist()}  max end {us[:,3].max():.2f} us; p50 end {np.median(us[:,3]):.2f}")


run("fwd L0 (lookup)", lambda: ops.dense_fwd2((None, None), (ut.w[0], it.w[0]), (ut.b[0], it.b[0]), (ut.acts[1], it.acts[1]), relu=True,
                                              lookups=tr._lookups(u, i, None)), 1024)
run("fwd L1", lambda: ops.dense_fwd2((ut.acts[1], it.acts[1]), (ut.w[1], it.w[1]), (ut.b[1], it.b[1]), (ut.acts[2], it.acts[2]), relu=False), 512)
none2 = (None, None)
run("bwd L1 dx only", lambda: ops.dense_bwd2((ut.acts[1], it.acts[1]), (ut.w[1], it.w[1]), (ut.dz[1], it.dz[1]), (ut.dz[0], it.dz[0]),
                                             (ut.acts[1], it.acts[1]), none2, none2), 1024)
run("bwd L1 dw only", lambda: ops.dense_bwd2((ut.acts[1], it.acts[1]), (ut.w[1], it.w[1]), (ut.dz[1], it.dz[1]), none2, none2,
                                             (ut.dw_slabs[1], it.dw_slabs[1]), (ut.db_slabs[1], it.db_slabs[1])), 512)
run("bwd L1 dx+dw", lambda: ops.dense_bwd2((ut.acts[1], it.acts[1]), (ut.w[1], it.w[1]), (ut.dz[1], it.dz[1]), (ut.dz[0], it.dz[0]),
                                           (ut.acts[1], it.acts[1]), (ut.dw_slabs[1], it.dw_slabs[1]), (ut.db_slabs[1], it.db_slabs[1])), 1536)
run("bwd L0 dx+dw (lookup)", lambda: ops.dense_bwd2((None, None), (ut.w[0], it.w[0]), (ut.dz[0], it.dz[0]), (ut.demb, it.demb),
                                                    none2, (ut.dw_slabs[0], it.dw_slabs[0]), (ut.db_slabs[0], it.db_slabs[0]),
                                                    lookups=tr._lookups(u, i, None)), 1024)
run("bwd L1 dx only, NO mask", lambda: ops.dense_bwd2((ut.acts[1], it.acts[1]), (ut.w[1], it.w[1]), (ut.dz[1], it.dz[1]), (ut.dz[0], it.dz[0]),
                                                      none2, none2, none2), 1024)
run("bwd L0 dx only", lambda: ops.dense_bwd2((ut.acts[0], it.acts[0]) if ut.acts[0] is not None else (ut.demb, it.demb), (ut.w[0], it.w[0]), (ut.dz[0], it.dz[0]), (ut.demb, it.demb),
                                             none2, none2, none2), 512)
