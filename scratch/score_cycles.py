"""Per-wave cycle totals of the scorer's tile loop (debug build -DTT_SCORE_STAMPS via TT_LIB_PATH): phases
0 GEMM1 / take X, 1 store S + epilogue, 2 prefetch issue, 3 GEMM2, 4 LDS store of the next tile (incl. its wait), 5 barrier."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_amazon_recommender_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
lib.tt_debug_score_cycles.restype = C.c_int
lib.tt_debug_score_cycles.argtypes = [C.c_void_p, C.c_int]
b, d = 8192, 128
q = torch.empty(b, d, device=dev); c = torch.empty(b, d, device=dev)
ops.fill_uniform_(q, 2, 1, -0.3, 0.6); ops.fill_uniform_(c, 2, 2, -0.3, 0.6)
ws = torch.empty(ops.retrieval_workspace_bytes(b, b, d), dtype=torch.uint8, device=dev)
lse = torch.empty(b, device=dev); pr = torch.empty(b, device=dev); loss = torch.empty(1, device=dev)
dq = torch.empty(b, d, device=dev); dc = torch.empty(b, d, device=dev)
for _ in range(5):
    ops.retrieval_fwd_bwd(q, c, 10.0, ws, lse, pr, loss, dq, dc)
torch.cuda.synchronize()
buf = np.zeros(2048 * 8 * 8, dtype=np.uint64)
assert lib.tt_debug_score_cycles(buf.ctypes.data, buf.size) == 0
a = buf.reshape(2048, 8, 8)[:512, :4].astype(np.float64)          # the LAST launch's values: pass 2 (BWD_S)
print("mode", a[0, 0, 7], "mean cycles per wave: GEMM1/X", a[..., 0].mean().round(), "S store + epilogue", a[..., 1].mean().round(),
      "prefetch issue", a[..., 2].mean().round(), "GEMM2", a[..., 3].mean().round(), "LDS store", a[..., 4].mean().round(),
      "barrier", a[..., 5].mean().round(), "| loop total", a[..., 6].mean().round())
