"""Phase stamps of optimizer_ids_kernel's sorting workgroups (debug build of sparse.hip with -DTT_SORT_STAMPS via TT_LIB_PATH):
0 start, 1 ids counted + appended, 2 barrier, 3 rows requested (r03: before the ranking), 4 (hot range only) list rebuilt, 5 ranked (sorted pairs in LDS), 6 rows updated."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_amazon_recommender_amd import _lib
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer

dev = torch.device("cuda:0")
lib = _lib.load()
lib.tt_debug_opt_stamps.restype = C.c_int
lib.tt_debug_opt_stamps.argtypes = [C.c_void_p, C.c_int]
cfg = TwoTowerConfig(n_users=5_000_000, n_items=10_000_000, embedding_dim=128, tower_dims=[256, 128], batch_size=8192)
tr = TwoTowerTrainer(cfg, dev, seed=3)
u, i = tr.synthetic_batch(3, 0)
for _ in range(5):
    tr.step(u, i)
torch.cuda.synchronize()
buf = np.zeros(1024 * 8, dtype=np.uint64)
assert lib.tt_debug_opt_stamps(buf.ctypes.data, buf.size) == 0
allw = buf.reshape(1024, 8).astype(np.int64)
s = allw[36:256]
s = s[s[:, 6] > s[:, 0]]
t0 = s[:, 0].min()
us = (s - t0) / 100.0
names = ["start", "appended", "barrier", "rows requested", "classified", "ranked", "updated", "ids landed"]
order = [0, 7, 4, 1, 2, 3, 5, 6]
print(len(s), "sorting WGs; mean stamp times (us):", ", ".join(f"{names[k]} {us[:, k].mean():.2f}" for k in order), "| max end", us[:, 6].max(), "start spread", us[:, 0].max())
pc = lambda a: " ".join(f"{np.percentile(a, q):.2f}" for q in (0, 10, 50, 90, 99, 100))
print("percentiles 0/10/50/90/99/100: ids landed", pc(us[:, 7]), "| barrier", pc(us[:, 2]), "| fast_apply done", pc(us[:, 3]), "| end", pc(us[:, 6]))
late = np.argsort(-us[:, 6])[:8]
print("latest WGs (index, start, barrier, fast_apply done, ranked, end):", [(int(k) + 36, *[round(float(us[k, c]), 2) for c in (0, 2, 3, 5, 6)]) for k in late])
d = allw[:36]
d = d[d[:, 6] > d[:, 0]]
ud = (d - t0) / 100.0
print(len(d), "dense blocks: start", pc(ud[:, 0]), "end", pc(ud[:, 6]), "ends:", ud[:, 6].round(1).tolist())
