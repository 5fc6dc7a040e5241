"""Round 4: layers 1 + 0 of the backward pass in one launch (tt_tower_bwd2_batched_f32) against one launch per layer:
bit-identity over several steps (Python sequence of launches, TT_COMPOSITE_STEP=0) and the error word; run under rocprofv3 for the durations."""
import os, sys
os.environ["TT_COMPOSITE_STEP"] = "0"
import torch
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd import ops
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer

dev = torch.device("cuda:0")
cfg = TwoTowerConfig(n_users=1_000_000, n_items=500_000, embedding_dim=128, tower_dims=[256, 128], temperature=0.1, l2_regularization=1e-6,
                     learning_rate=0.001, optimizer="sgd", batch_size=8192, dropout_rate=float(os.environ.get("DROP", "0")))
a = TwoTowerTrainer(cfg, dev, seed=3)
b = TwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=3)
b.bwd2_ws = ops.tower_bwd2_workspace(cfg.batch_size, dev)
steps = int(os.environ.get("STEPS", "30"))
for s in range(steps):
    u, i = a.synthetic_batch(3, s, "Z" if s % 2 else "U")
    la = a.step(u, i).clone()
    lb = b.step(u, i).clone()
    assert torch.equal(la, lb), (s, la.item(), lb.item())
torch.cuda.synchronize()
err = b.bwd2_ws.view(torch.int32)[4 * (cfg.batch_size // 64)].item()
same = torch.equal(a.user_table, b.user_table) and torch.equal(a.item_table, b.item_table) and torch.equal(a.dense_flat, b.dense_flat)
print("steps", steps, "identical", same, "wait ran out", err, "counters zero", int(b.bwd2_ws.view(torch.int32).abs().sum().item()) == 0)
assert same and err == 0
