"""timing only: the fused two-layer backward launch - variants via TT_LIB_PATH, durations from rocprofv3"""
import os, sys
os.environ["TT_COMPOSITE_STEP"] = "0"
import torch
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd import ops
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer
dev = torch.device("cuda:0")
cfg = TwoTowerConfig(n_users=1_000_000, n_items=500_000, embedding_dim=128, tower_dims=[256, 128], temperature=0.1, l2_regularization=1e-6,
                     learning_rate=0.001, optimizer="sgd", batch_size=8192)
b = TwoTowerTrainer(cfg, dev, seed=3)
if os.environ.get("FUSED", "1") == "1":
    b.bwd2_ws = ops.tower_bwd2_workspace(cfg.batch_size, dev)
for s in range(60):
    u, i = b.synthetic_batch(3, s, "U")
    b.step(u, i)
torch.cuda.synchronize()
