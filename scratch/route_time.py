"""Duration of the multi-table route kernel (2 tables, 8192 ids each) at world 1/2/8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from two_tower_amazon_recommender_amd import ops, _lib
dev = torch.device("cuda:0")
for world in (1, 2, 8):
    n = 8192
    cap = n if world == 1 else 2048 if world == 8 else 8192
    ids = [torch.randint(0, 10_000_000, (n,), device=dev) for _ in range(2)]
    send = torch.empty(world * 2 * cap, dtype=torch.int64, device=dev)
    pos = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(2)]
    fl = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.profile_enable("route", 256)
    for _ in range(50):
        ops.route_tables_by_owner(ids, world, [10_000_000] * 2, [0, 5_000_000], cap, send, pos, fl)
    ms, seen = _lib.profile_read("route")
    print(world, "route us", 1e3 * sum(ms[10:]) / len(ms[10:]), fl.tolist(), flush=True)
