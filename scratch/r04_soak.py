"""Soak: N train steps (composite entry) on a small skewed problem; prints a digest of every table / dense parameter at the end.
Run once with the forward lookup's row-range id lists (default) and once with TT_ID_BUCKETS=0 (the optimizer's own scan): the
digests must be equal - counters reset, generations, overflow fallbacks and the ranked path's offsets over thousands of steps."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
opt = sys.argv[2] if len(sys.argv) > 2 else "adagrad"
dev = torch.device("cuda:0")
cfg = TwoTowerConfig(n_users=200_000, n_items=50_000, embedding_dim=128, tower_dims=[256, 128], batch_size=4096, optimizer=opt,
                     learning_rate=0.01, dropout_rate=0.1)
tr = TwoTowerTrainer(cfg, dev, seed=11)
tr.flag_poll_every = int(os.environ.get("POLL", "50"))
# PHASES=1: long runs of power-law batches, then uniform ones (the skew probe switches the optimizer path back and forth);
# default: the two kinds interleaved (the probe sees whichever batch it lands on)
phases = os.environ.get("PHASES", "0") != "0"
switches, was = 0, tr.one_launch_optimizer(cfg.batch_size)
for s in range(steps):
    kind = ("Z" if (s // 137) % 2 else "U") if phases else ("Z" if s % 3 else "U")
    u, i = tr.synthetic_batch(11, s, kind)
    tr.step(u, i)
    now = tr.one_launch_optimizer(cfg.batch_size)
    switches += now != was
    was = now
torch.cuda.synchronize()
tr.check_ids()
h = hashlib.sha256()
for t in (tr.user_table, tr.item_table, tr.dense_flat) + ((tr.user_accum, tr.item_accum, tr.dense_accum) if opt == "adagrad" else ()):
    h.update(t.cpu().numpy().tobytes())
print(f"steps {steps} opt {opt} lists {os.environ.get('TT_ID_BUCKETS', '1')} skew_limit {tr.skew_limit} path switches {switches} loss {tr.loss.item():.6f} digest {h.hexdigest()}")
