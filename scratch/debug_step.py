import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import synth, two_tower as tt
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer
dev = torch.device("cuda:0")
n_users, n_items, dim, tower_dims, batch = 50_000, 100_000, 128, [256, 128], 8192
seed = 1001
cfg = TwoTowerConfig(n_users=n_users, n_items=n_items, embedding_dim=dim, tower_dims=tower_dims, temperature=0.1,
                     l2_regularization=1e-6, learning_rate=0.001, optimizer="sgd", batch_size=batch)
tr = TwoTowerTrainer(cfg, dev, seed=seed)
ref = tt.synthetic_state(seed, n_users, n_items, dim, tower_dims, dtype=np.float64)
def e(name, got, want):
    got = got.cpu().numpy(); d = np.abs(got - want)
    rows = np.unique(np.where(d > 1e-3 * np.abs(want).max())[0])
    print(f"  {name:10s} err {d.max():.3e} ref {np.abs(want).max():.3e} bad rows {len(rows)} {rows[:8]}")
for step in range(3):
    uid = synth.batch_ids(seed, 3, step, batch, n_users, "U"); iid = synth.batch_ids(seed, 4, step, batch, n_items, "U")
    du, di = tr.synthetic_batch(seed, step, "U")
    # before the step: compare tables at the rows used
    loss = tr.step(du, di).item()
    r = tt.forward_backward(ref, uid, iid, temperature=0.1, l2=1e-6)
    print("step", step, loss, r["loss"])
    e("q", tr.user_tower.acts[-1], r["q"]); e("c", tr.item_tower.acts[-1], r["c"])
    e("lse", tr.lse, r["lse"])
    e("dq", tr.user_tower.dz[-1], r["dq"]); e("dc", tr.item_tower.dz[-1], r["dc"])
    e("due", tr.user_tower.demb, r["due"]); e("die", tr.item_tower.demb, r["die"])
    e("uh1", tr.user_tower.acts[1], r["user_acts"][1])
    tt.train_step(ref, uid, iid, lr=0.001, optimizer="sgd", temperature=0.1, l2=1e-6)
    e("utable", tr.user_table, ref.user_table); e("itable", tr.item_table, ref.item_table)
    e("uw0", tr.user_tower.w[0], ref.user_tower.weights[0]); e("ub0", tr.user_tower.b[0], ref.user_tower.biases[0])
    e("uw1", tr.user_tower.w[1], ref.user_tower.weights[1]); e("ub1", tr.user_tower.b[1], ref.user_tower.biases[1])
