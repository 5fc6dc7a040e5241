"""One-GPU measurements for DESIGN.md section 6's per-GPU time budget of cfg4 at N = 8 (a MODEL): the per-GPU shapes of the
8-way sharded step - B_local = 2048 queries against the 16384 all-gathered candidates, towers and embedding traffic at 2048
rows - timed with the library's own hipEvent brackets (tt_profile_*) over repeated calls."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_amazon_recommender_amd import _lib, ops  # noqa: E402
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer  # noqa: E402

dev = torch.device("cuda:0")
out = {}


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


# ---- scorer: this rank's 2048 queries x the global batch's 16384 candidates, positives at diag_offset = rank * 2048
bq, bc, d = 2048, 16384, 128
q = torch.empty(bq, d, device=dev); c = torch.empty(bc, d, device=dev)
ops.fill_uniform_(q, 2, 1, -0.3, 0.6); ops.fill_uniform_(c, 2, 2, -0.3, 0.6)
ws = torch.empty(ops.retrieval_workspace_bytes(bq, bc, d), dtype=torch.uint8, device=dev)
lse = torch.empty(bq, device=dev); pr = torch.empty(bq, device=dev); loss = torch.empty(1, device=dev)
dq = torch.empty(bq, d, device=dev); dc = torch.empty(bc, d, device=dev)
for prec in ("f32", "bf16x3"):
    out[f"scorer_2048x16384_{prec}_us"] = timed(lambda: ops.retrieval_fwd_bwd(q, c, 10.0, ws, lse, pr, loss, dq, dc, diag_offset=3 * bq, precision=prec))
# ---- towers + optimizer at 2048 rows per GPU: the plain trainer's launches (same kernels as the sharded step's), per tag
cfg = TwoTowerConfig(n_users=5_000_000 // 8, n_items=100_000_000 // 8, embedding_dim=128, tower_dims=[256, 128], batch_size=2048)
tr = TwoTowerTrainer(cfg, dev, seed=3)
u, i = tr.synthetic_batch(3, 0)
for _ in range(10):
    tr.step(u, i)
torch.cuda.synchronize()
tags = "dense_fwd,dense_bwd,optimizer,score_fused,score_bwd,score_aux"
_lib.profile_enable(tags, 256)
for _ in range(40):
    tr.step(u, i)
torch.cuda.synchronize()
for t in tags.split(","):
    v = _lib.profile_read(t, 256)[0]
    out[f"plain_b2048_{t}_us_per_step"] = sum(v) / 40 * 1e3
_lib.profile_enable("")
out["plain_b2048_step_us"] = timed(lambda: tr.step(u, i), 100)
# ---- the exchange's kernels at world 8: routing of 2 x 2048 ids, owner gather / requester scatter of 4096 rows, owner update
world, cap = 8, 512
ids = [u, i]
send = torch.empty(world * 2 * cap, dtype=torch.int64, device=dev)
pos = [torch.empty(2048, dtype=torch.int64, device=dev) for _ in range(2)]
flags = torch.zeros(2, dtype=torch.int32, device=dev)
rows = [cfg.n_users * 8, cfg.n_items * 8]
offs = [0, (rows[0] + world - 1) // world]
out["route_2x2048_w8_us"] = timed(lambda: ops.route_tables_by_owner(ids, world, rows, offs, cap, send, pos, flags))
shard = torch.empty(offs[1] + (rows[1] + world - 1) // world, d, device=dev)
ops.fill_uniform_(shard, 1, 1, -0.05, 0.1)
recv = torch.full((world * 2 * cap,), -1, dtype=torch.int64, device=dev)
n_valid = 4096
valid = torch.randint(0, shard.shape[0], (n_valid,), device=dev)
recv[torch.randperm(world * 2 * cap, device=dev)[:n_valid]] = valid
rows_out = torch.empty(world * 2 * cap, d, device=dev)
out["owner_gather_8192slots_us"] = timed(lambda: ops.embedding_gather(shard, recv, out=rows_out))
grads = torch.empty(4096, d, device=dev); ops.fill_uniform_(grads, 1, 9, -1.0, 2.0)
pf = torch.cat(pos)
out["scatter_rows_4096_us"] = timed(lambda: ops.scatter_rows(grads, pf, rows_out))
plan = ops.SparsePlan(world * 2 * cap, dev)
out["owner_optimizer_8192slots_us"] = timed(lambda: ops.optimizer_step_ids_("sgd", [(shard, None, rows_out, recv, plan)], tr._segs, 1e-6))
print(json.dumps(out))
