"""Phase stamps of part_sort_kernel (debug build -DTT_SORT_STAMPS via TT_LIB_PATH): 0 start, 1 ids counted, 2 barrier,
3 scan done, 4 compaction done (barrier), 5 rank sort written."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_amazon_recommender_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
lib.tt_debug_sort_stamps.restype = C.c_int
lib.tt_debug_sort_stamps.argtypes = [C.c_void_p, C.c_int]
n, rows = 8192, 5_000_000
ids = torch.empty(n, dtype=torch.int64, device=dev)
ops.fill_ids_(ids, 1, 3, rows, "U")
pl = ops.SparsePlan(n, dev)
x = torch.randn(4096, 4096, device=dev)
for hot in (False, True):
    for _ in range(5):
        if hot:
            for _ in range(20):
                x @ x
        pl.run(ids, rows)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert lib.tt_debug_sort_stamps(buf.ctypes.data, buf.size) == 0
    s = buf.reshape(1024, 8)[:64, :6].astype(np.int64)
    s = s[s[:, 5] > s[:, 0]]
    t0 = s[:, 0].min()
    us = (s - t0) / 100.0
    print("hot" if hot else "cold", len(s), "WGs; mean stamp times (us):", us.mean(0).round(2).tolist(), "max end", us[:, 5].max())
