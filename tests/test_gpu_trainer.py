"""Whole train step on the GPU (HIP kernels through the C ABI) vs the f64 oracle, same synthetic
state and batches.  Parity unpinned by the reference (oracle/__init__.py)."""
import numpy as np
import pytest
import torch

from oracle import synth, two_tower as tt
from two_tower_amazon_recommender_amd import ops
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer

pytestmark = pytest.mark.gpu


def make(dev, n_users, n_items, dim, tower_dims, batch, opt, seed, l2=1e-6, dropout=0.0):
    cfg = TwoTowerConfig(n_users=n_users, n_items=n_items, embedding_dim=dim, tower_dims=tower_dims, temperature=0.1,
                         l2_regularization=l2, learning_rate=0.001, optimizer=opt, batch_size=batch, dropout_rate=dropout)
    tr = TwoTowerTrainer(cfg, dev, seed=seed)
    ref = tt.synthetic_state(seed, n_users, n_items, dim, tower_dims, dtype=np.float64, optimizer=opt)
    return cfg, tr, ref


def test_synthetic_init_is_bit_identical_to_oracle(dev):
    cfg, tr, _ = make(dev, 3000, 2000, 64, [128, 64], 256, "sgd", 77)
    ref32 = tt.synthetic_state(77, 3000, 2000, 64, [128, 64], dtype=np.float32)
    assert np.array_equal(tr.user_table.cpu().numpy(), ref32.user_table)
    assert np.array_equal(tr.item_table.cpu().numpy(), ref32.item_table)
    for t, (tower, rt) in enumerate(((tr.user_tower, ref32.user_tower), (tr.item_tower, ref32.item_tower))):
        for l in range(2):
            assert np.array_equal(tower.w[l].cpu().numpy(), rt.weights[l]), (t, l)
            assert not tower.b[l].any()


@pytest.mark.parametrize("name,shape,opt,variant", [
    ("cfg1", (10_000, 10_000, 32, [32], 256), "sgd", "U"),
    ("cfg1z", (10_000, 10_000, 32, [32], 256), "adagrad", "Z"),
    ("cfg2-small", (100_000, 100_000, 64, [64], 4096), "sgd", "Z"),
    ("cfg3-small", (50_000, 100_000, 128, [256, 128], 8192), "sgd", "U"),
    ("cfg3-small-adagrad", (50_000, 100_000, 128, [256, 128], 2048), "adagrad", "Z"),
    ("ref-config-towers", (5_000, 5_000, 128, [512, 256, 128], 1024), "sgd", "U"),
    ("ref-config-dropout0.1", (5_000, 5_000, 128, [512, 256, 128], 1024), "adagrad", "Z"),
    ("ragged-batch-777", (3_000, 2_000, 64, [96, 64], 777), "adagrad", "Z"),
    ("cfg5-dims-256", (4_000, 4_000, 256, [512, 256], 1536), "adagrad", "U"),
    # the same bars with the scorer's matrix products in the f32-emulated bf16x3 precision (split-bf16 on the bf16 MFMA)
    ("cfg3-small-bf16x3", (50_000, 100_000, 128, [256, 128], 8192), "sgd", "U"),
    ("ref-config-dropout0.1-bf16x3", (5_000, 5_000, 128, [512, 256, 128], 1024), "adagrad", "Z"),
    ("cfg5-dims-256-bf16x3", (4_000, 4_000, 256, [512, 256], 1536), "adagrad", "U"),
])
def test_train_steps_match_oracle(dev, name, shape, opt, variant):
    n_users, n_items, dim, tower_dims, batch = shape
    seed = 1001
    rate = 0.1 if "dropout" in name else 0.0          # configs/data_config.yaml:58
    cfg, tr, ref = make(dev, n_users, n_items, dim, tower_dims, batch, opt, seed, dropout=rate)
    if name.endswith("bf16x3"):
        cfg.scorer_precision = "bf16x3"
        cfg.validate()
    for step in range(3):
        uid = synth.batch_ids(seed, synth.TID_USER_IDS, step, batch, n_users, variant)
        iid = synth.batch_ids(seed, synth.TID_ITEM_IDS, step, batch, n_items, variant)
        du, di = tr.synthetic_batch(seed, step, variant)
        assert np.array_equal(du.cpu().numpy(), uid) and np.array_equal(di.cpu().numpy(), iid)   # bit-exact indices
        loss = tr.step(du, di).item()
        # ReLU's derivative is discontinuous at 0: hand the oracle the masks the device used (a hidden
        # pre-activation within f32 rounding of 0 may have the other sign in f64); everything else is f64.
        masks = tuple([(t.acts[l + 1] > 0).cpu().numpy() for l in range(len(tower_dims) - 1)]
                      for t in (tr.user_tower, tr.item_tower))
        drop = None
        if rate:
            dims = [dim] + tower_dims
            per_tower = [[synth.dropout_keep(seed, synth.dropout_tid(t, l), step * batch, batch, dims[l + 1], rate)
                          for l in range(len(tower_dims) - 1)] for t in (0, 1)]
            drop = (per_tower[0], per_tower[1], per_tower[0][0][1])
        r = tt.train_step(ref, uid, iid, lr=0.001, optimizer=opt, temperature=0.1, l2=1e-6, relu_masks=masks, dropout=drop)
        tr.check_ids()
        # loss: |d|/B <= 1e-4 and relative <= 1e-4
        assert abs(loss - r["loss"]) / batch <= 1e-4 and abs(loss - r["loss"]) <= 1e-4 * abs(r["loss"]), (step, loss, r["loss"])
        # embedding-row gradients that fed the sparse update
        for got, want in ((tr.user_tower.demb, r["due"]), (tr.item_tower.demb, r["die"])):
            err = np.abs(got.cpu().numpy() - want).max()
            assert err <= 1e-4 * np.abs(want).max(), (step, err, np.abs(want).max())
    # state after 3 steps
    for got, want in ((tr.user_table, ref.user_table), (tr.item_table, ref.item_table)):
        g = got.cpu().numpy()
        assert np.abs(g - want).max() <= 2e-6, np.abs(g - want).max()
    for tower, rt in ((tr.user_tower, ref.user_tower), (tr.item_tower, ref.item_tower)):
        for l in range(len(tower_dims)):
            # 3 updates of size lr*|g| ~ 1e-3 * O(1..30): a wrong gradient shows as >= 1e-4; f32 sums over
            # 8192 rows leave ~3e-6 (measured)
            assert np.abs(tower.w[l].cpu().numpy() - rt.weights[l]).max() <= 1e-5
            assert np.abs(tower.b[l].cpu().numpy() - rt.biases[l]).max() <= 1e-5
    if opt == "adagrad":
        assert np.abs(tr.user_accum.cpu().numpy() - ref.user_accum).max() <= 1e-4 * ref.user_accum.max()


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_hashed_category_feature_matches_oracle(dev, opt):
    """BASELINE configs[4]: a 30-bucket hashed category table whose row is ADDED to the item embedding before the item
    tower.  Three steps against the f64 oracle: loss, the shared input gradient, and the 30 rows after ~270
    duplicate-gradient sums per row and step (sorted-slot pieces of 64 + the finish kernel)."""
    n_users, n_items, dim, tower_dims, batch, nb, seed = 3000, 2000, 64, [128, 64], 1024, 30, 1005
    cfg = TwoTowerConfig(n_users=n_users, n_items=n_items, embedding_dim=dim, tower_dims=tower_dims, temperature=0.1,
                         l2_regularization=1e-6, learning_rate=0.001, optimizer=opt, batch_size=batch, n_category_buckets=nb)
    tr = TwoTowerTrainer(cfg, dev, seed=seed)
    ref = tt.synthetic_state(seed, n_users, n_items, dim, tower_dims, dtype=np.float64, optimizer=opt, n_category_buckets=nb)
    assert np.array_equal(tr.cat_table.cpu().numpy(), ref.cat_table.astype(np.float32))
    with pytest.raises(ValueError):
        tr.step(*tr.synthetic_batch(seed, 0, "Z"))                     # the model has the feature: ids are required
    for step in range(3):
        du, di = tr.synthetic_batch(seed, step, "Z")
        dc = tr.synthetic_categories(seed, step)
        cid = synth.batch_ids(seed, synth.TID_CATEGORY_IDS, step, batch, nb, "Z")
        assert np.array_equal(dc.cpu().numpy(), cid)
        loss = tr.step(du, di, category_ids=dc).item()
        masks = tuple([(t.acts[l + 1] > 0).cpu().numpy() for l in range(len(tower_dims) - 1)]
                      for t in (tr.user_tower, tr.item_tower))
        r = tt.train_step(ref, du.cpu().numpy(), di.cpu().numpy(), lr=0.001, optimizer=opt, temperature=0.1, l2=1e-6,
                          relu_masks=masks, category_ids=cid)
        tr.check_ids()
        assert abs(loss - r["loss"]) / batch <= 1e-4 and abs(loss - r["loss"]) <= 1e-4 * abs(r["loss"]), (step, loss, r["loss"])
        err = np.abs(tr.item_tower.demb.cpu().numpy() - r["die"]).max()
        assert err <= 1e-4 * np.abs(r["die"]).max()
    for got, want in ((tr.cat_table, ref.cat_table), (tr.item_table, ref.item_table), (tr.user_table, ref.user_table)):
        assert np.abs(got.cpu().numpy() - want).max() <= 2e-6
    assert (tr.cat_table.cpu().numpy() != synth.embedding_table(seed, synth.TID_CATEGORY_TABLE, nb, dim)).mean() > 0.9
    if opt == "adagrad":
        assert np.abs(tr.cat_accum.cpu().numpy() - ref.cat_accum).max() <= 1e-4 * ref.cat_accum.max()
    # validation loss and the item corpus use the feature too
    val = tr.evaluate(du, di, category_ids=dc).item()
    rv = tt.forward_backward(ref, du.cpu().numpy(), di.cpu().numpy(), temperature=0.1, category_ids=cid)
    assert abs(val - rv["loss"]) <= 1e-4 * abs(rv["loss"])
    item_cat = torch.arange(n_items, device=dev) % nb
    corpus = tr.item_corpus_embeddings(item_cat)
    want = tt.tower_fwd(ref.item_table + ref.cat_table[item_cat.cpu().numpy()], ref.item_tower.weights, ref.item_tower.biases)[-1]
    assert np.abs(corpus.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max()
    # checkpoint round trip keeps the table; graph replay takes the ids
    sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in tr.state_dict().items()}
    tr2 = TwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=seed + 1)
    tr2.load_state_dict(sd)
    assert torch.equal(tr2.cat_table, tr.cat_table)
    l_eager = tr.step(du, di, category_ids=dc).clone()
    tr2.capture_graph()
    l_graph = tr2.step_graph(du, di, dc).clone()
    assert torch.equal(l_eager, l_graph) and torch.equal(tr.cat_table, tr2.cat_table)


def test_full_size_train_step_properties_cfg3(dev):
    """BASELINE configs[2] at full size (5M users x 10M items x 128, towers 256-128, batch 8192), properties that
    need no O(B^2) host work: only the batch's rows change, the first loss is ~ln(B) (near-uniform softmax at init),
    and the whole step is bit-reproducible from the seed."""
    shape = (5_000_000, 10_000_000, 128, [256, 128], 8192)
    sums = []
    for rep in range(2):
        cfg = TwoTowerConfig(n_users=shape[0], n_items=shape[1], embedding_dim=shape[2], tower_dims=shape[3], temperature=0.1,
                             l2_regularization=1e-6, learning_rate=0.001, optimizer="sgd", batch_size=shape[4])
        tr = TwoTowerTrainer(cfg, dev, seed=1003)
        before_u, before_i = tr.user_table.clone(), tr.item_table.clone()
        dense_before = tr.dense_flat.clone()
        u, i = tr.synthetic_batch(1003, 0, "Z")
        loss = tr.step(u, i).item()
        tr.check_ids()
        assert abs(loss / shape[4] - np.log(shape[4])) < 0.05
        for table, before, ids in ((tr.user_table, before_u, u), (tr.item_table, before_i, i)):
            changed = (table != before).any(dim=1).nonzero().flatten()
            uniq = torch.unique(ids)
            assert torch.isin(changed, uniq).all(), "a row outside the batch was modified"
            assert changed.numel() >= 0.99 * uniq.numel()
            assert torch.isfinite(table[uniq]).all()
        assert (tr.dense_flat != dense_before).float().mean().item() > 0.5
        loss2 = tr.step(*tr.synthetic_batch(1003, 1, "Z")).item()
        sums.append((loss, loss2, tr.user_table.view(torch.int32).sum(dtype=torch.int64).item(),
                     tr.item_table.view(torch.int32).sum(dtype=torch.int64).item(),
                     tr.dense_flat.view(torch.int32).sum(dtype=torch.int64).item()))
        del tr, before_u, before_i
        torch.cuda.empty_cache()
    assert sums[0] == sums[1], "the train step is not bit-reproducible"


@pytest.mark.parametrize("engine", ["plain", "sharded_world1_rccl"])
def test_full_size_train_step_properties_cfg4(dev, engine):
    """BASELINE configs[3] at FULL size on one GPU: 5M users x 100M items x 128 (53.8 GB of tables), towers 256-128, batch
    16384, SGD - un-sharded through TwoTowerTrainer, and through ShardedTwoTowerTrainer on a one-rank "nccl" (= RCCL) group
    with every collective really issued (route, id / row / gradient all-to-alls, dense all-reduce, owner update in one
    launch where the received list fits: 2 x 16384 ids here, so sort plan + apply).  Properties that need no O(B^2) host work and no second copy of
    the tables: the batch's rows change and sampled rows outside the batch do not, the first loss is ~ln(B), two runs from
    the seed agree bit for bit - and the two engines agree with each other bit for bit."""
    import os
    import torch.distributed as dist
    free, total_mem = torch.cuda.mem_get_info()
    if total_mem < 120e9:
        pytest.skip("needs > 120 GB of HBM")
    nu, ni, d, dims, b = 5_000_000, 100_000_000, 128, [256, 128], 16384
    if engine != "plain":
        assert not dist.is_initialized()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = "29581"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        sums = []
        for rep in range(2):
            cfg = TwoTowerConfig(n_users=nu, n_items=ni, embedding_dim=d, tower_dims=dims, temperature=0.1, l2_regularization=1e-6,
                                 learning_rate=0.001, optimizer="sgd", batch_size=b)
            if engine == "plain":
                tr = TwoTowerTrainer(cfg, dev, seed=1004)
            else:
                from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer
                tr = ShardedTwoTowerTrainer(cfg, dev, seed=1004, negatives="global", force_collectives=True)
                assert tr.collectives
            u, i = tr.synthetic_batch(1004, 0, "Z")
            probe = {}
            for name, table, ids, rows in (("user", tr.user_table, u, nu), ("item", tr.item_table, i, ni)):
                uniq = torch.unique(ids)
                other = torch.randint(0, rows, (200_000,), device=dev, generator=torch.Generator(device=dev).manual_seed(rep))
                other = other[~torch.isin(other, uniq)]
                probe[name] = (uniq, other, table[uniq].clone(), table[other].clone())
            dense_before = tr.dense_flat.clone()
            loss = tr.step(u, i).item()
            tr.check_ids()
            assert abs(loss / b - np.log(b)) < 0.05, loss / b
            for name, table in (("user", tr.user_table), ("item", tr.item_table)):
                uniq, other, rows_before, other_before = probe[name]
                assert torch.equal(table[other], other_before), f"{name}: a row outside the batch was modified"
                assert (table[uniq] != rows_before).any(dim=1).float().mean().item() >= 0.99
                assert torch.isfinite(table[uniq]).all()
            assert (tr.dense_flat != dense_before).float().mean().item() > 0.5
            loss2 = tr.step(*tr.synthetic_batch(1004, 1, "Z")).item()
            tr.check_ids()
            sums.append((loss, loss2, tr.user_table[probe["user"][0]].view(torch.int32).sum(dtype=torch.int64).item(),
                         tr.item_table[probe["item"][0]].view(torch.int32).sum(dtype=torch.int64).item(),
                         tr.dense_flat.view(torch.int32).sum(dtype=torch.int64).item()))
            del tr, probe, table, uniq, other, rows_before, other_before
            torch.cuda.empty_cache()
        assert sums[0] == sums[1], "the cfg4 train step is not bit-reproducible"
        # both engines must land on the same numbers (one rank: every collective is the identity)
        ref = getattr(test_full_size_train_step_properties_cfg4, "_sums", None)
        if ref is not None:
            assert ref == sums[0], "plain and sharded(world 1, RCCL) cfg4 steps differ"
        test_full_size_train_step_properties_cfg4._sums = sums[0]
    finally:
        if engine != "plain":
            dist.destroy_process_group()


def test_full_size_train_step_properties_cfg5(dev):
    """BASELINE configs[4] at FULL size on one GPU: 54M users x 48M items x 256 (+ Adagrad accumulators: 209 GB of
    tables), towers 256->512->256, batch 32768, fused sparse Adagrad, 30-bucket hashed category feature.  Properties
    that need no O(B^2) host work and no second copy of the tables: the batch's rows (and their accumulators) change
    and sampled rows outside the batch do not, the first loss is ~ln(B), and the step is bit-reproducible from the seed."""
    free, total_mem = torch.cuda.mem_get_info()
    if total_mem < 250e9:
        pytest.skip("needs a 288 GB MI355X")
    nu, ni, d, dims, b, nb = 54_000_000, 48_000_000, 256, [512, 256], 32768, 30
    sums = []
    for rep in range(2):
        cfg = TwoTowerConfig(n_users=nu, n_items=ni, embedding_dim=d, tower_dims=dims, temperature=0.1, l2_regularization=1e-6,
                             learning_rate=0.001, optimizer="adagrad", batch_size=b, n_category_buckets=nb)
        tr = TwoTowerTrainer(cfg, dev, seed=1005)
        u, i = tr.synthetic_batch(1005, 0, "Z")
        c = tr.synthetic_categories(1005, 0)
        probe = {}
        for name, table, accum, ids, rows in (("user", tr.user_table, tr.user_accum, u, nu), ("item", tr.item_table, tr.item_accum, i, ni)):
            uniq = torch.unique(ids)
            other = torch.randint(0, rows, (200_000,), device=dev, generator=torch.Generator(device=dev).manual_seed(rep))
            other = other[~torch.isin(other, uniq)]
            probe[name] = (uniq, other, table[uniq].clone(), table[other].clone())
        cat_before, dense_before = tr.cat_table.clone(), tr.dense_flat.clone()
        loss = tr.step(u, i, category_ids=c).item()
        tr.check_ids()
        assert abs(loss / b - np.log(b)) < 0.05, loss / b
        for name, table, accum in (("user", tr.user_table, tr.user_accum), ("item", tr.item_table, tr.item_accum)):
            uniq, other, rows_before, other_before = probe[name]
            assert torch.equal(table[other], other_before), f"{name}: a row outside the batch was modified"
            assert torch.equal(accum[other], torch.full_like(other_before, 0.1))
            changed = (table[uniq] != rows_before).any(dim=1)
            assert changed.float().mean().item() >= 0.99
            assert (accum[uniq] >= 0.1).all() and (accum[uniq] > 0.1).any(dim=1).float().mean().item() >= 0.99
            assert torch.isfinite(table[uniq]).all()
        assert (tr.cat_table != cat_before).any() and (tr.dense_flat != dense_before).float().mean().item() > 0.5
        loss2 = tr.step(*tr.synthetic_batch(1005, 1, "Z"), category_ids=tr.synthetic_categories(1005, 1)).item()
        sums.append((loss, loss2, tr.user_table[probe["user"][0]].view(torch.int32).sum(dtype=torch.int64).item(),
                     tr.item_table[probe["item"][0]].view(torch.int32).sum(dtype=torch.int64).item(),
                     tr.item_accum[probe["item"][0]].view(torch.int32).sum(dtype=torch.int64).item(),
                     tr.cat_table.view(torch.int32).sum(dtype=torch.int64).item(),
                     tr.dense_flat.view(torch.int32).sum(dtype=torch.int64).item()))
        del tr, probe, table, accum, uniq, other, rows_before, other_before      # (loop variables keep 49 GB tables alive)
        torch.cuda.empty_cache()
    assert sums[0] == sums[1], "the cfg5 train step is not bit-reproducible"


def test_wrong_length_batches_are_refused(dev):
    """The tower buffers hold exactly batch_size rows: a longer batch would write past them, a shorter one would leave
    stale rows that the scorer still reads — both raise before anything is enqueued (evaluate, evaluate_topk and the
    option vectors too)."""
    from two_tower_amazon_recommender_amd.metrics import FactorizedTopK
    cfg, tr, _ = make(dev, 500, 400, 32, [32], 256, "sgd", 5)
    u, i = tr.synthetic_batch(5, 0)
    for bad_u, bad_i in ((u[:200], i[:200]), (torch.cat([u, u]), torch.cat([i, i]))):
        with pytest.raises(ValueError):
            tr.step(bad_u, bad_i)
        with pytest.raises(ValueError):
            tr.evaluate(bad_u, bad_i)
        with pytest.raises(ValueError):
            tr.evaluate_topk(bad_u, bad_i, FactorizedTopK(ks=(1,)), corpus=torch.zeros(400, 32, device=dev))
    with pytest.raises(ValueError):
        tr.step(u, i, sample_weight=torch.ones(255, device=dev))
    with pytest.raises(ValueError):
        tr.evaluate(u, i, candidate_ids=i[:100])
    # ops level: the kernels index outputs / option vectors by n_ids / nq / nc without looking at their shapes
    with pytest.raises(RuntimeError):
        ops.embedding_gather(tr.user_table, u, out=torch.empty(128, 32, device=dev))
    with pytest.raises(RuntimeError):
        ops.embedding_gather2(tr.user_table, u, torch.empty(256, 32, device=dev), tr.item_table, i, torch.empty(255, 32, device=dev))
    q = torch.zeros(64, 32, device=dev)
    ws = torch.empty(ops.retrieval_workspace_bytes(64, 64, 32), dtype=torch.uint8, device=dev)
    v = lambda n: torch.empty(n, device=dev)
    with pytest.raises(RuntimeError):
        ops.retrieval_fwd(q, q, 1.0, ws, v(64), v(64), v(1), sample_weight=v(63))
    with pytest.raises(RuntimeError):
        ops.retrieval_fwd_bwd(q, q, 1.0, ws, v(64), v(64), v(1), torch.empty(64, 32, device=dev), torch.empty(63, 32, device=dev))
    tr.step(u, i)                # and the trainer still works afterwards
    tr.check_ids()


def test_checkpoint_resume_continues_the_dropout_stream(dev):
    """state_dict carries step_index and dropout_seed: after a resume the counter-based dropout masks continue where the
    checkpoint stopped (no replayed masks), so resumed training is bit-identical to uninterrupted training."""
    def fresh():
        return make(dev, 800, 700, 32, [64, 32], 256, "adagrad", 17, dropout=0.2)[1]
    a = fresh()
    for s in range(3):
        a.step(*a.synthetic_batch(17, s))
    sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in a.state_dict().items()}
    assert sd["step_index"] == 3 and sd["dropout_seed"] == 17
    for s in range(3, 5):
        a.step(*a.synthetic_batch(17, s))
    b2 = fresh()
    b2.load_state_dict(sd)
    assert b2.step_index == 3
    for s in range(3, 5):
        b2.step(*b2.synthetic_batch(17, s))
    assert torch.equal(a.user_table, b2.user_table) and torch.equal(a.item_table, b2.item_table)
    assert torch.equal(a.dense_flat, b2.dense_flat) and torch.equal(a.dense_accum, b2.dense_accum)


def test_out_of_range_id_is_reported_by_the_periodic_poll_with_its_step(dev):
    """No per-epoch wait: step() polls the flag every `flag_poll_every` steps through an asynchronous 4-byte copy and
    raises one interval later, naming the step range."""
    cfg, tr, _ = make(dev, 100, 100, 32, [32], 256, "sgd", 5)
    tr.flag_poll_every = 2
    u, i = tr.synthetic_batch(5, 0)
    bad = u.clone()
    bad[7] = 100
    tr.step(u, i)                                   # step 0: poll starts a (clean) copy
    tr.step(bad, i)                                 # step 1: the bad id
    with pytest.raises(IndexError, match="step"):
        for _ in range(6):                          # polls at steps 2, 4, 6: copy the raised flag, then see it
            tr.step(u, i)
            torch.cuda.synchronize()
    tr.step(u, i)
    tr.check_ids()                                  # flag was cleared by the raise


def test_skew_probe_moves_the_steps_to_the_plan_path_and_back_without_changing_a_bit(dev):
    """r04: ids that crowd a few row ranges (ids ~ rows * u^4: a vocabulary in order of frequency) make ONE workgroup of the
    one-launch optimizer sort and apply a third of the batch.  The trainer's probe (tt_id_range_load every `flag_poll_every`
    steps, read from pinned memory when it has landed - never waited for) sees the overloaded range and the following steps
    take plan + optimizer step; uniform batches bring the one-launch form back.  Both paths are bit-identical, so a trainer
    that never switches (skew_limit 0) must end in exactly the same state."""
    cfg = TwoTowerConfig(n_users=2_000_000, n_items=1_000_000, embedding_dim=64, tower_dims=[128, 64], temperature=0.1,
                         l2_regularization=1e-6, learning_rate=0.001, optimizer="adagrad", batch_size=8192)
    a = TwoTowerTrainer(cfg, dev, seed=41)
    b = TwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=41)
    b.skew_limit = 0
    a.flag_poll_every = 3
    assert a.skew_limit == 384 and a.one_launch_optimizer(cfg.batch_size)
    paths = []
    for step in range(16):
        variant = "Z" if step < 8 else "U"
        u, i = a.synthetic_batch(41, step, variant)
        la = a.step(u, i).clone()
        torch.cuda.synchronize()                      # (the test's own wait, so that "has landed" is deterministic)
        paths.append(a.one_launch_optimizer(cfg.batch_size))
        lb = b.step(u, i).clone()
        assert torch.equal(la, lb), step
    # probe at step 0 (power-law ids) -> read at step 1: plan path from step 1 on; probe at step 9 (uniform) -> back at step 10
    assert paths[0] is True and not any(paths[1:9]) and all(paths[10:]), paths
    assert a.range_load < 288 and b.range_load == 0
    assert torch.equal(a.user_table, b.user_table) and torch.equal(a.item_table, b.item_table)
    assert torch.equal(a.user_accum, b.user_accum) and torch.equal(a.dense_flat, b.dense_flat)


def test_out_of_range_id_is_reported(dev):
    cfg, tr, _ = make(dev, 100, 100, 32, [32], 256, "sgd", 5)
    u, i = tr.synthetic_batch(5, 0)
    u[7] = 100
    tr.step(u, i)
    with pytest.raises(IndexError):
        tr.check_ids()


def test_loss_decreases_over_steps(dev):
    cfg, tr, _ = make(dev, 2000, 2000, 64, [64], 1024, "adagrad", 9)
    u, i = tr.synthetic_batch(9, 0)
    first = tr.step(u, i).item()
    for _ in range(20):
        last = tr.step(u, i).item()
    assert last < first


@pytest.mark.parametrize("opt,variant,buckets", [("sgd", "U", 0), ("adagrad", "Z", 30)])
def test_fused_launches_equal_the_separate_ones_bit_for_bit(dev, opt, variant, buckets):
    """The train step's fusions change launches, not arithmetic: lookup inside the layer-0 GEMM loaders, ReLU sign bits as
    the dx mask, sort + sparse apply + dense update in ONE optimizer launch from the raw ids — against gather launch,
    plan launch, sparse launch(es) and dense launch.  Loss, tables, accumulators and dense weights after several steps
    must be identical bit for bit (uniform / Zipf ids, with and without the hashed category table)."""
    cfg = TwoTowerConfig(n_users=5000, n_items=3000, embedding_dim=64, tower_dims=[128, 64], temperature=0.1,
                         l2_regularization=1e-6, learning_rate=0.001, optimizer=opt, batch_size=2048,
                         n_category_buckets=buckets)
    a = TwoTowerTrainer(cfg, dev, seed=37)
    b = TwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=37)
    b.fuse_sort = False
    b.fuse_optimizer = False
    b.fuse_lookup = False
    assert a.fuse_sort and a.fuse_optimizer and a.fuse_lookup
    for step in range(4):
        u, i = a.synthetic_batch(37, step, variant)
        extra = {"category_ids": a.synthetic_categories(37, step)} if buckets else {}
        la = a.step(u, i, **extra).clone()
        lb = b.step(u, i, **extra).clone()
        assert torch.equal(la, lb), step
    a.check_ids(); b.check_ids()
    assert torch.equal(a.user_table, b.user_table) and torch.equal(a.item_table, b.item_table)
    assert torch.equal(a.dense_flat, b.dense_flat)
    if buckets:
        assert torch.equal(a.cat_table, b.cat_table)
    if opt == "adagrad":
        assert torch.equal(a.user_accum, b.user_accum) and torch.equal(a.item_accum, b.item_accum)
    # the middle form too: one optimizer launch, but behind a separate plan launch
    c = TwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=37)
    c.fuse_sort = False
    for step in range(4):
        u, i = c.synthetic_batch(37, step, variant)
        extra = {"category_ids": c.synthetic_categories(37, step)} if buckets else {}
        c.step(u, i, **extra)
    assert torch.equal(a.user_table, c.user_table) and torch.equal(a.item_table, c.item_table) and torch.equal(a.dense_flat, c.dense_flat)


def test_two_layer_backward_in_one_launch_is_bit_identical_and_leaves_its_counters_zeroed(dev):
    """tt_tower_bwd2_batched_f32 (r04, OFF by default: it measures slower, profiles/r04_bwd2_ab.txt): layers 1 and 0 of both towers'
    backward pass in one launch, the lower layer's tiles waiting inside the launch (agent-scope release / acquire on a counter per
    64-row block) for the rows of dz they read - against one launch per layer, with dropout and the hashed category table, uniform
    and power-law ids: every step's loss and the final state identical bit for bit, no wait ran out, counters back at zero."""
    cfg = TwoTowerConfig(n_users=50_000, n_items=30_000, embedding_dim=64, tower_dims=[128, 64], temperature=0.1, l2_regularization=1e-6,
                         learning_rate=0.001, optimizer="adagrad", batch_size=2048, dropout_rate=0.1, n_category_buckets=30)
    a = TwoTowerTrainer(cfg, dev, seed=47)
    b = TwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=47)
    a.use_composite = b.use_composite = False
    assert ops.tower_bwd2_supported(cfg.batch_size, 64, 128, 64) and not ops.tower_bwd2_supported(1000, 64, 128, 64)
    b.bwd2_ws = ops.tower_bwd2_workspace(cfg.batch_size, dev)
    for step in range(6):
        u, i = a.synthetic_batch(47, step, "Z" if step % 2 else "U")
        c = a.synthetic_categories(47, step)
        la = a.step(u, i, category_ids=c).clone()
        lb = b.step(u, i, category_ids=c).clone()
        assert torch.equal(la, lb), step
    words = b.bwd2_ws.view(torch.int32)
    assert int(words.abs().sum().item()) == 0           # counters zeroed by the last consumer, error word never set
    assert torch.equal(a.user_table, b.user_table) and torch.equal(a.item_table, b.item_table) and torch.equal(a.cat_table, b.cat_table)
    assert torch.equal(a.dense_flat, b.dense_flat) and torch.equal(a.dense_accum, b.dense_accum)


@pytest.mark.parametrize("opt,buckets,precision", [("sgd", 0, "f32"), ("adagrad", 30, "f32"), ("sgd", 0, "bf16x3")])
def test_composite_step_entry_equals_the_python_sequence_bit_for_bit(dev, opt, buckets, precision):
    """tt_train_step_f32 (one C call per step: the nine launches enqueued in C) against the same launches issued one by one
    from Python, with everything the step takes: dropout (the per-step counter), sample weights, sampling-probability
    correction, accidental-hit ids, the hashed category table, Adagrad, the bf16x3 scorer."""
    cfg = TwoTowerConfig(n_users=5000, n_items=3000, embedding_dim=64, tower_dims=[128, 128], temperature=0.1,
                         l2_regularization=1e-6, learning_rate=0.01, optimizer=opt, batch_size=1024, dropout_rate=0.1,
                         n_category_buckets=buckets, scorer_precision=precision)
    a = TwoTowerTrainer(cfg, dev, seed=41)
    b = TwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=41)
    b.use_composite = False
    assert a.use_composite
    g = torch.Generator(device="cpu").manual_seed(3)
    for step in range(4):
        if step == 3:          # the precision may be switched between steps (bench.py's second line does): the C struct follows
            other = "bf16x3" if precision == "f32" else "f32"
            a.cfg.scorer_precision = b.cfg.scorer_precision = other
        u, i = a.synthetic_batch(41, step, "Z")
        extra = {"category_ids": a.synthetic_categories(41, step)} if buckets else {}
        if step >= 1:
            extra["sample_weight"] = (0.5 + torch.rand(1024, generator=g)).to(dev)
        if step >= 2:
            extra["candidate_sampling_probability"] = (0.001 + 0.3 * torch.rand(1024, generator=g)).to(dev)
            extra["candidate_ids"] = i
        la = a.step(u, i, **extra).clone()
        lb = b.step(u, i, **extra).clone()
        assert torch.equal(la, lb), step
    assert a._cstep is not None and b._cstep is None
    a.check_ids(); b.check_ids()
    assert torch.equal(a.user_table, b.user_table) and torch.equal(a.item_table, b.item_table)
    assert torch.equal(a.dense_flat, b.dense_flat) and torch.equal(a.per_row, b.per_row)
    if buckets:
        assert torch.equal(a.cat_table, b.cat_table)
    if opt == "adagrad":
        assert torch.equal(a.user_accum, b.user_accum) and torch.equal(a.dense_accum, b.dense_accum)


def test_sharded_owner_side_takes_the_plan_path_under_skew_and_stays_bit_identical(dev):
    """r04: the owner side of the row-sharded step has the same probe as the plain trainer (ShardedTables._poll_skew on the
    RECEIVED ids): power-law batches crowd the first row ranges of the combined shard, the probe sees it one lookup later and
    the owner update takes plan() + apply(); uniform batches bring apply_ids back.  Bit-identical to the plain trainer
    throughout (which switches by its own probe on the same batches)."""
    from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer
    cfg = TwoTowerConfig(n_users=2_000_000, n_items=1_000_000, embedding_dim=64, tower_dims=[128, 64], temperature=0.1,
                         l2_regularization=1e-6, learning_rate=0.001, optimizer="adagrad", batch_size=8192)
    tr = TwoTowerTrainer(cfg, dev, seed=43)
    sh = ShardedTwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=43)
    sh.emb.probe_every = tr.flag_poll_every = 3
    assert sh.emb.fused_apply and sh.emb.probe_segs is not None
    seen = []
    for step in range(12):
        variant = "Z" if step < 6 else "U"
        u, i = tr.synthetic_batch(43, step, variant)
        l1 = tr.step(u, i).clone()
        l2 = sh.step(u.clone(), i.clone()).clone()
        torch.cuda.synchronize()
        seen.append(sh.emb._fused_now)
        assert torch.equal(l1, l2), step
    assert seen[0] is True and not any(seen[1:7]) and all(seen[8:]), seen
    assert sh.emb.range_load < 288
    assert torch.equal(sh.user_table, tr.user_table) and torch.equal(sh.item_table, tr.item_table)
    assert torch.equal(sh.dense_flat, tr.dense_flat)


@pytest.mark.parametrize("opt,variant", [("sgd", "U"), ("adagrad", "Z")])
def test_sharded_trainer_world1_is_bit_identical_to_single_gpu_trainer(dev, opt, variant):
    """The row-sharded step (route / de-dup / exchange buffers / owner update) with one rank must reproduce the
    plain trainer bit for bit: same kernels, same summation order."""
    from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer
    shape = (5000, 3000, 64, [128, 64], 1024)
    cfg, tr, _ = make(dev, *shape, opt, 31)
    cfg2 = TwoTowerConfig(**cfg.__dict__)
    sh = ShardedTwoTowerTrainer(cfg2, dev, seed=31)
    assert torch.equal(sh.user_table, tr.user_table) and torch.equal(sh.dense_flat, tr.dense_flat)
    for step in range(3):
        u, i = tr.synthetic_batch(31, step, variant)
        u2, i2 = sh.synthetic_batch(31, step, variant)
        assert torch.equal(u, u2) and torch.equal(i, i2)
        l1 = tr.step(u, i).clone()
        l2 = sh.step(u2, i2).clone()
        assert torch.equal(l1, l2)
    sh.check_ids()
    assert torch.equal(sh.user_table, tr.user_table) and torch.equal(sh.item_table, tr.item_table)
    assert torch.equal(sh.dense_flat, tr.dense_flat)
    if opt == "adagrad":
        assert torch.equal(sh.emb.accum_shard(0), tr.user_accum) and torch.equal(sh.emb.accum_shard(1), tr.item_accum)


def test_retrieval_task_autograd_matches_oracle(dev):
    from two_tower_amazon_recommender_amd.tasks import Retrieval
    nq, d = 300, 64
    q = synth.uniform_f32(51, 1, nq * d, -0.3, 0.6).reshape(nq, d)
    c = synth.uniform_f32(51, 2, nq * d, -0.3, 0.6).reshape(nq, d)
    w = synth.uniform_f32(51, 3, nq, 0.5, 1.5)
    p = synth.uniform_f32(51, 4, nq, 0.01, 0.3)
    ids = synth.ids_powerlaw(51, 5, nq, 40)
    tq = torch.from_numpy(q).to(dev).requires_grad_()
    tc = torch.from_numpy(c).to(dev).requires_grad_()
    task = Retrieval(temperature=0.1, remove_accidental_hits=True)
    loss = task(tq, tc, sample_weight=torch.from_numpy(w).to(dev), candidate_sampling_probability=torch.from_numpy(p).to(dev),
                candidate_ids=torch.from_numpy(ids).to(dev))
    (0.5 * loss).backward()
    kw = dict(temperature=0.1, sample_weight=w, candidate_sampling_probability=p, candidate_ids=ids, remove_accidental_hits=True)
    rl, _, _ = tt.retrieval_loss(q, c, **kw)
    rdq, rdc = tt.retrieval_grad(q, c, **kw)
    assert abs(loss.item() - rl) <= 1e-4 * abs(rl)
    assert np.abs(tq.grad.cpu().numpy() - 0.5 * rdq).max() <= 1e-4 * np.abs(rdq).max()
    assert np.abs(tc.grad.cpu().numpy() - 0.5 * rdc).max() <= 1e-4 * np.abs(rdc).max()
    # no temperature, rectangular candidates
    task2 = Retrieval()
    c2 = torch.from_numpy(np.concatenate([c, c[:50]])).to(dev)
    l2 = task2(tq.detach(), c2)
    assert abs(l2.item() - tt.retrieval_loss(q, np.concatenate([c, c[:50]]))[0]) <= 1e-4 * abs(l2.item())


def test_train_cli_synthetic_runs_and_learns(dev, tmp_path):
    from two_tower_amazon_recommender_amd import train
    cfgp = tmp_path / "cfg.yaml"
    cfgp.write_text("model:\n  embedding_dim: 32\n  user_tower_dims: [64, 32]\n  item_tower_dims: [64, 32]\n"
                    "  dropout_rate: 0.0\n  l2_regularization: 1e-6\n  training:\n    batch_size: 512\n    learning_rate: 0.05\n"
                    "    epochs: 3\n    patience: 5\n    validation_freq: 1\n  retrieval:\n    candidate_sampling: in_batch\n"
                    "    temperature: 0.1\n    top_k_eval: [1, 10]\n")
    ck = tmp_path / "ck.pt"
    import contextlib, io, json
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        rc = train.main(["--config", str(cfgp), "--synthetic", "40000", "--synthetic-users", "2000", "--synthetic-items", "1500",
                         "--save", str(ck)])
    assert rc == 0 and ck.exists()
    res = json.loads(out.getvalue().strip().splitlines()[-1])
    hist = res["history"]
    assert hist[-1]["train_loss_per_pair"] < hist[0]["train_loss_per_pair"]
    # top_k_eval of the YAML (configs/data_config.yaml:71): recall/ndcg of the held-out pairs against the whole corpus
    vm = res["val_metrics"]
    assert set(vm) == {"recall@1", "ndcg@1", "recall@10", "ndcg@10"}
    assert 0.0 <= vm["recall@1"] <= vm["recall@10"] <= 1.0 and vm["ndcg@10"] <= vm["recall@10"] + 1e-12
    # (no "better than random" bar: the synthetic pairs are drawn independently, and uncorrected in-batch negatives
    # push popular items DOWN — the known bias candidate_sampling_probability exists to remove; with the correction
    # the model can at least learn popularity, so recall@10 must improve on the uncorrected run)
    out2 = io.StringIO()
    with contextlib.redirect_stdout(out2):
        assert train.main(["--config", str(cfgp), "--synthetic", "40000", "--synthetic-users", "2000", "--synthetic-items", "1500",
                           "--correct-sampling-bias"]) == 0
    vm2 = json.loads(out2.getvalue().strip().splitlines()[-1])["val_metrics"]
    assert vm2["recall@10"] > vm["recall@10"]
    sd = torch.load(ck, weights_only=True)
    assert sd["user_table"].shape == (2000, 32)


def test_train_cli_distributed_one_rank_equals_single_gpu(dev, tmp_path):
    """`train-model --distributed` (the torchrun path: row-sharded trainer, RCCL group, per-rank data slice, all-reduced
    losses and metric tallies, per-rank checkpoint) with ONE rank reproduces the single-GPU run's numbers exactly."""
    import contextlib, io, json
    import torch.distributed as dist
    from two_tower_amazon_recommender_amd import train
    assert not dist.is_initialized()
    cfgp = tmp_path / "cfg.yaml"
    cfgp.write_text("model:\n  embedding_dim: 32\n  user_tower_dims: [64, 32]\n  item_tower_dims: [64, 32]\n"
                    "  dropout_rate: 0.1\n  l2_regularization: 1e-6\n  training:\n    batch_size: 512\n    learning_rate: 0.05\n"
                    "    epochs: 2\n    patience: 5\n    validation_freq: 1\n  retrieval:\n    candidate_sampling: in_batch\n"
                    "    temperature: 0.1\n    top_k_eval: [1, 10, 100]\n")
    common = ["--config", str(cfgp), "--synthetic", "30000", "--synthetic-users", "2000", "--synthetic-items", "1500",
              "--category-buckets", "30", "--correct-sampling-bias"]
    outs = []
    for extra in ([], ["--distributed", "--save", str(tmp_path / "ck.pt")]):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            assert train.main(common + extra) == 0
        outs.append(json.loads(buf.getvalue().strip().splitlines()[-1]))
    assert not dist.is_initialized()
    for a, b in zip(outs[0]["history"], outs[1]["history"]):
        assert a["train_loss_per_pair"] == b["train_loss_per_pair"] and a["val_loss_per_pair"] == b["val_loss_per_pair"]
    assert outs[0]["val_metrics"] == outs[1]["val_metrics"]
    sd = torch.load(tmp_path / "ck.pt.rank0of1", weights_only=True)
    assert sd["user_shard"].shape == (2000, 32) and sd["world"] == 1


def test_train_cli_parquet_with_hashed_category_column(dev, tmp_path):
    """train-model on a parquet shaped like prepare_training_data.py:216-218's output, with its `category` column
    hashed into 30 buckets on the GPU and fed to the item tower (BASELINE configs[4])."""
    import pandas as pd
    from oracle import hashing
    from two_tower_amazon_recommender_amd import data as datamod, train
    rng = np.random.default_rng(1)
    n, cats = 6000, ["All_Beauty", "Books", "Electronics", "Home_and_Kitchen", "Toys_and_Games", None]
    df = pd.DataFrame({"user_idx": rng.integers(0, 800, n), "item_idx": rng.integers(0, 600, n),
                       "category": [cats[k] for k in rng.integers(0, len(cats), n)], "rating": 5.0})
    df.loc[0, ["user_idx", "item_idx"]] = [799, 599]
    p = tmp_path / "combined_interactions.parquet"
    df.to_parquet(p, compression="snappy", index=False)
    codes, values = datamod.read_category_values(p)
    got = datamod.category_buckets(codes, values, 30, dev)
    want = hashing.hash_buckets(["Unknown" if c is None else c for c in df["category"]], 30)
    assert np.array_equal(got, want)                                 # GPU hash == oracle, row by row
    cfgp = tmp_path / "cfg.yaml"
    cfgp.write_text("model:\n  embedding_dim: 32\n  user_tower_dims: [32]\n  item_tower_dims: [32]\n"
                    "  dropout_rate: 0.0\n  l2_regularization: 1e-6\n  training:\n    batch_size: 256\n    learning_rate: 0.05\n"
                    "    epochs: 2\n    patience: 5\n    validation_freq: 1\n  retrieval:\n    candidate_sampling: in_batch\n"
                    "    temperature: 0.1\n    top_k_eval: [1, 10]\n")
    ck = tmp_path / "ck.pt"
    import contextlib, io, json
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        assert train.main(["--config", str(cfgp), "--data", str(p), "--category-buckets", "30", "--save", str(ck)]) == 0
    assert "recall@10" in json.loads(out.getvalue().strip().splitlines()[-1])["val_metrics"]   # corpus built with item categories
    sd = torch.load(ck, weights_only=True)
    assert sd["cat_table"].shape == (30, 32) and sd["config"]["n_category_buckets"] == 30
    touched = np.unique(want)
    init = synth.embedding_table(42, synth.TID_CATEGORY_TABLE, 30, 32)
    moved = np.flatnonzero((sd["cat_table"].cpu().numpy() != init).any(axis=1))
    assert set(moved) == set(touched.tolist())                       # exactly the buckets that occur were trained


def test_evaluate_topk_against_item_corpus(dev):
    from two_tower_amazon_recommender_amd.metrics import FactorizedTopK
    cfg, tr, ref = make(dev, 3000, 2500, 64, [64], 512, "adagrad", 13)
    u, i = tr.synthetic_batch(13, 0, "Z")
    for _ in range(30):
        tr.step(u, i)                                               # memorise one batch
    corpus = tr.item_corpus_embeddings()
    assert corpus.shape == (2500, 64)
    m = FactorizedTopK(ks=(1, 10, 100), temperature=0.1)
    rank = tr.evaluate_topk(u, i, m, corpus)
    # oracle: same tables/weights pulled from the device, f64 towers
    ut, it = tr.user_table.cpu().numpy().astype(np.float64), tr.item_table.cpu().numpy().astype(np.float64)
    q = tt.tower_fwd(ut[u.cpu().numpy()], [w.cpu().numpy().astype(np.float64) for w in tr.user_tower.w],
                     [b.cpu().numpy().astype(np.float64) for b in tr.user_tower.b])[-1]
    c = tt.tower_fwd(it, [w.cpu().numpy().astype(np.float64) for w in tr.item_tower.w],
                     [b.cpu().numpy().astype(np.float64) for b in tr.item_tower.b])[-1]
    assert np.abs(corpus.cpu().numpy() - c).max() <= 1e-5
    lo, hi = tt.retrieval_rank_bounds(q, c, i.cpu().numpy(), temperature=0.1, eps=1e-4)
    r = rank.cpu().numpy()
    assert (r >= lo).all() and (r <= hi).all()
    res = m.result()
    assert res["recall@100"] > 0.3                                  # trained pairs are retrievable


def test_two_tower_model_facade(dev):
    from two_tower_amazon_recommender_amd.model import TwoTowerModel
    cfg = TwoTowerConfig(n_users=500, n_items=400, embedding_dim=32, tower_dims=[32], batch_size=128, optimizer="sgd")
    m = TwoTowerModel(cfg, dev, seed=3)
    u, i = m.trainer.synthetic_batch(3, 0)
    # tfrs.Model.train_step's dict: total_loss = loss + the Dense kernels' L2 terms, evaluated on the weights the step started from
    l2_before = float(m.trainer.l2_penalty().item())
    out = m.train_step({"user_idx": u, "item_idx": i})
    assert out["loss"].item() > 0
    assert l2_before > 0 and abs(out["regularization_loss"].item() - l2_before) <= 1e-5 * l2_before
    assert abs(out["total_loss"].item() - (out["loss"].item() + l2_before)) <= 1e-5 * abs(out["total_loss"].item())
    plain = m.train_step({"user_idx": u, "item_idx": i}, report_regularization=False)
    assert plain["regularization_loss"] is None and torch.equal(plain["total_loss"], plain["loss"])
    val = m.test_step({"user_id_encoded": u, "item_id_encoded": i})      # the preprocessor's column names work too
    assert val["loss"].item() > 0
    with pytest.raises(KeyError):
        m.train_step({"user": u, "item": i})
    # with the hashed category feature: raw strings (hashed on the GPU) or ready bucket ids
    from oracle import hashing
    cfg2 = TwoTowerConfig(n_users=500, n_items=400, embedding_dim=32, tower_dims=[32], batch_size=128, optimizer="sgd",
                          n_category_buckets=30)
    a, b = TwoTowerModel(cfg2, dev, seed=3), TwoTowerModel(TwoTowerConfig(**cfg2.__dict__), dev, seed=3)
    cats = [("Books", "Electronics", "All_Beauty")[k % 3] for k in range(128)]
    la = a.train_step({"user_idx": u, "item_idx": i, "category": cats})["loss"]
    lb = b.train_step({"user_idx": u, "item_idx": i,
                       "category_bucket": torch.from_numpy(hashing.hash_buckets(cats, 30)).to(dev)})["loss"]
    assert torch.equal(la, lb) and torch.equal(a.trainer.cat_table, b.trainer.cat_table)
    with pytest.raises(KeyError):
        a.train_step({"user_idx": u, "item_idx": i})


def test_retrieval_task_validates_in_its_training_precision_and_reuses_one_workspace(dev):
    """Retrieval(precision="bf16x3") under no_grad runs the validation op in bf16x3 too (r03: it silently validated in f32), and
    the custom ops' scorer workspace is kept at the LARGEST size seen per (kind, device, stream) - a loop alternating two batch
    shapes does not re-allocate - until release_workspaces()."""
    from two_tower_amazon_recommender_amd import torch_ops
    from two_tower_amazon_recommender_amd.tasks import Retrieval
    torch_ops.release_workspaces()
    q = torch.from_numpy(synth.uniform_f32(91, 1, 1024 * 128, -0.3, 0.6).reshape(1024, 128)).to(dev)
    c = torch.from_numpy(synth.uniform_f32(91, 2, 1024 * 128, -0.3, 0.6).reshape(1024, 128)).to(dev)
    t32, tbx = Retrieval(temperature=0.1), Retrieval(temperature=0.1, precision="bf16x3")
    with torch.no_grad():
        v32, vbx = t32(q, c), tbx(q, c)
    direct = torch.ops.twotower.retrieval_loss_value(q, c, None, None, None, 10.0, 0, 0, "bf16x3")[0]
    assert torch.equal(vbx, direct)                                  # the task passed its precision on
    assert abs(vbx.item() - v32.item()) <= 1e-4 * abs(v32.item())    # same bar as the training form
    qg = q.clone().requires_grad_(True)
    assert abs(tbx(qg, c).item() - vbx.item()) <= 1e-5 * abs(vbx.item())
    big = [b for b in torch_ops._WS_CACHE.values()]
    ptrs = {b.data_ptr() for b in big}
    with torch.no_grad():
        for n in (256, 1024, 512, 1024, 256):
            t32(q[:n], c[:n])
    assert {b.data_ptr() for b in torch_ops._WS_CACHE.values()} == ptrs      # no re-allocation for the smaller shapes
    torch_ops.release_workspaces()
    assert not torch_ops._WS_CACHE and not torch_ops._PLAN_CACHE


@pytest.mark.parametrize("negatives,nb", [("local", 0), ("global", 0), ("local", 30)])
def test_sharded_trainer_one_rank_rccl_collectives(dev, negatives, nb):
    """The N>1 code path with its collectives really issued — asynchronous all_to_all_single (ids int64, rows f32,
    gradient rows), the dense all-reduce, and for negatives="global" all_gather_into_tensor + reduce_scatter_tensor —
    on a one-rank "nccl" (= RCCL) group, which is all one GPU allows.  Everything must equal the plain trainer bit for
    bit (a one-rank sum is the identity)."""
    import os
    import torch.distributed as dist
    from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer
    assert not dist.is_initialized()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = "29578"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        cfg = TwoTowerConfig(n_users=5000, n_items=3000, embedding_dim=64, tower_dims=[128, 64], temperature=0.1,
                             l2_regularization=1e-6, learning_rate=0.001, optimizer="adagrad", batch_size=1024,
                             n_category_buckets=nb)
        tr = TwoTowerTrainer(cfg, dev, seed=37)
        sh = ShardedTwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=37, negatives=negatives, force_collectives=True)
        assert sh.collectives and sh.emb.recv_ids.data_ptr() != sh.emb.send_ids.data_ptr()
        batches = [tr.synthetic_batch(37, step, "Z") for step in range(4)]
        for step in range(4):
            u, i = batches[step]
            kw = {"category_ids": tr.synthetic_categories(37, step)} if nb else {}
            l1 = tr.step(u, i, **kw).clone()
            # steps 0-1 hand the next step's ids over (routed beside the scorer; step 0 also issues their id all-to-all early)
            l2 = sh.step(u, i, next_ids=batches[step + 1] if step < 2 else None, prefetch_exchange=(step == 0), **kw).clone()
            assert torch.equal(l1, l2)
        sh.check_ids()
        assert torch.equal(sh.user_table, tr.user_table) and torch.equal(sh.item_table, tr.item_table)
        n_dense = tr.dense_flat.numel()
        assert torch.equal(sh.dense_flat[:n_dense], tr.dense_flat)
        assert torch.equal(sh.emb.accum_shard(0), tr.user_accum)
        if nb:      # replicated table, gradient through the all-reduce bucket + dense update == the sparse update, bit for bit
            assert torch.equal(sh.cat_table, tr.cat_table) and torch.equal(sh.cat_accum, tr.cat_accum)
        # checkpoint of this rank's shard: save -> fresh trainer -> load -> the next step is bit-identical
        import io
        buf = io.BytesIO()
        torch.save(sh.state_dict(), buf)
        buf.seek(0)
        sh2 = ShardedTwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=99, negatives=negatives, force_collectives=True)
        sh2.load_state_dict(torch.load(buf, weights_only=True))
        u, i = tr.synthetic_batch(37, 7, "Z")
        kw = {"category_ids": tr.synthetic_categories(37, 7)} if nb else {}
        la, lb = sh.step(u, i, **kw).clone(), sh2.step(u, i, **kw).clone()
        assert torch.equal(la, lb) and torch.equal(sh.emb.table, sh2.emb.table) and torch.equal(sh.dense_flat, sh2.dense_flat)
        bad = sh.state_dict()
        bad["world"] = 8
        with pytest.raises(ValueError):
            sh2.load_state_dict(bad)
    finally:
        dist.destroy_process_group()


def test_sharded_trainer_global_negatives_world1_equals_local(dev):
    """negatives="global" (all-gather candidates, diag offset, reduce-scatter dC) degenerates to the local form on one
    rank; needs a process group, so a single-rank gloo group is created on the fly when none exists."""
    import torch.distributed as dist
    from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer
    created = False
    if not dist.is_initialized():
        import os
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        cfg = TwoTowerConfig(n_users=4000, n_items=3000, embedding_dim=64, tower_dims=[64], batch_size=512, optimizer="sgd",
                             dropout_rate=0.1)
        a = ShardedTwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=17, negatives="local")
        b = ShardedTwoTowerTrainer(TwoTowerConfig(**cfg.__dict__), dev, seed=17, negatives="global")
        for step in range(2):
            u, i = a.synthetic_batch(17, step, "Z")
            la = a.step(u, i).clone(); lb = b.step(u, i).clone()
            assert torch.equal(la, lb)
        assert torch.equal(a.emb.table, b.emb.table) and torch.equal(a.dense_flat, b.dense_flat)
        a.check_ids(); b.check_ids()
    finally:
        if created:
            dist.destroy_process_group()


def test_asymmetric_towers_match_oracle(dev):
    """user_tower_dims != item_tower_dims (separate keys in configs/data_config.yaml:56-57): per-tower launches."""
    n_users, n_items, dim, batch, seed = 3000, 2000, 64, 512, 23
    cfg = TwoTowerConfig(n_users=n_users, n_items=n_items, embedding_dim=dim, tower_dims=[128, 64], item_tower_dims=[96, 80, 64],
                         temperature=0.1, l2_regularization=1e-6, learning_rate=0.001, optimizer="adagrad", batch_size=batch)
    tr = TwoTowerTrainer(cfg, dev, seed=seed)
    ref = tt.synthetic_state(seed, n_users, n_items, dim, [128, 64], dtype=np.float64, optimizer="adagrad", item_tower_dims=[96, 80, 64])
    for step in range(2):
        u, i = tr.synthetic_batch(seed, step, "Z")
        loss = tr.step(u, i).item()
        masks = tuple([(t.acts[l + 1] > 0).cpu().numpy() for l in range(t.n_layers - 1)] for t in (tr.user_tower, tr.item_tower))
        r = tt.train_step(ref, u.cpu().numpy(), i.cpu().numpy(), lr=0.001, optimizer="adagrad", temperature=0.1, l2=1e-6,
                          relu_masks=masks)
        assert abs(loss - r["loss"]) <= 1e-4 * abs(r["loss"])
        assert np.abs(tr.item_tower.demb.cpu().numpy() - r["die"]).max() <= 1e-4 * np.abs(r["die"]).max()
    assert np.abs(tr.item_tower.w[1].cpu().numpy() - ref.item_tower.weights[1]).max() <= 1e-5
    assert np.abs(tr.user_table.cpu().numpy() - ref.user_table).max() <= 2e-6


def test_custom_ops_pass_opcheck_and_match_the_oracle(dev):
    """torch.library.opcheck on every registered op (schema, fake implementation, autograd registration, AOT dispatch),
    then numbers: the autograd path through torch.ops.twotower.retrieval_loss / dense_fwd against the f64 oracle."""
    from two_tower_amazon_recommender_amd import torch_ops  # noqa: F401
    g = torch.Generator(device="cpu").manual_seed(3)
    b, d, n = 192, 64, 128
    q = (torch.rand(b, d, generator=g) * 0.6 - 0.3).to(dev).requires_grad_()
    c = (torch.rand(b + 40, d, generator=g) * 0.6 - 0.3).to(dev).requires_grad_()
    w = (torch.rand(b, generator=g) + 0.5).to(dev)
    p = (torch.rand(b + 40, generator=g) * 0.5 + 0.001).to(dev)
    ids = torch.randint(0, 50, (b + 40,), generator=g).to(dev)
    opc = torch.library.opcheck
    opc(torch.ops.twotower.retrieval_loss, (q, c, w, p, ids, 10.0, 7, 0))
    opc(torch.ops.twotower.retrieval_loss, (q.detach(), c.detach(), None, None, None, 1.0, 0, 5))
    opc(torch.ops.twotower.retrieval_loss_value, (q.detach(), c.detach(), w, None, None, 10.0, 0, 0))
    opc(torch.ops.twotower.retrieval_rank, (q.detach(), c.detach(), torch.arange(b, device=dev), p, 10.0))
    opc(torch.ops.twotower.retrieval_batch_rank, (q.detach(), c.detach(), p, ids, 10.0, 0))
    table = torch.rand(500, d, generator=g).to(dev)
    opc(torch.ops.twotower.embedding_gather, (table, torch.randint(0, 500, (77,), generator=g).to(dev)))
    x = (torch.rand(b, d, generator=g) - 0.5).to(dev).requires_grad_()
    wt = (torch.rand(d, n, generator=g) - 0.5).to(dev).requires_grad_()
    bias = (torch.rand(n, generator=g) - 0.5).to(dev).requires_grad_()
    opc(torch.ops.twotower.dense_fwd, (x, wt, bias, True))
    opc(torch.ops.twotower.dense_bwd, (x.detach(), wt.detach(), torch.rand(b, n, generator=g).to(dev), None))
    acc = torch.full_like(table, 0.1)
    grads = torch.rand(77, d, generator=g).to(dev)
    opc(torch.ops.twotower.sparse_update_, (table.clone(), acc, grads, torch.randint(0, 500, (77,), generator=g).to(dev), "adagrad", 0.01, 1e-7))
    # numbers: loss(dense(x)) differentiated by autograd through the two custom ops
    y = torch.ops.twotower.dense_fwd(x, wt, bias, True)
    c2 = (torch.rand(b + 40, n, generator=g) * 0.6 - 0.3).to(dev).requires_grad_()
    loss, _, _, _ = torch.ops.twotower.retrieval_loss(y, c2, w, p, ids, 10.0, 7, 0)
    (loss * 0.5).backward()
    xn, wn, bn, cn = (t.detach().cpu().double().numpy() for t in (x, wt, bias, c2))
    yn = np.maximum(xn @ wn + bn, 0)
    kw = dict(temperature=0.1, sample_weight=w.cpu().numpy(), candidate_sampling_probability=p.cpu().numpy(),
              candidate_ids=ids.cpu().numpy(), remove_accidental_hits=True, diag_offset=7)
    rl, _, _ = tt.retrieval_loss(yn, cn, **kw)
    dyn, dcn = tt.retrieval_grad(yn, cn, **kw)
    dyn, dcn = 0.5 * dyn * (yn > 0), 0.5 * dcn
    assert abs(loss.item() - rl) <= 1e-4 * abs(rl)
    for got, ref in ((x.grad, dyn @ wn.T), (wt.grad, xn.T @ dyn), (bias.grad, dyn.sum(0)), (c2.grad, dcn)):
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max()


def test_retrieval_task_with_factorized_topk_and_loss_metrics(dev):
    """tfrs.tasks.Retrieval(metrics=FactorizedTopK(candidates=...)): the task updates the metric when compute_metrics
    is true, given every candidate's index in the corpus."""
    from two_tower_amazon_recommender_amd.metrics import FactorizedTopK
    from two_tower_amazon_recommender_amd.tasks import Retrieval
    g = torch.Generator(device="cpu").manual_seed(5)
    corpus = (torch.rand(1000, 32, generator=g) - 0.5).to(dev)
    idx = torch.randperm(1000, generator=g)[:128].to(dev)
    q = (corpus[idx] + 0.01 * torch.rand(128, 32, generator=g).to(dev)).requires_grad_()      # queries near their true items

    class Mean:
        def __init__(self):
            self.v = []

        def update_state(self, x):
            self.v.append(float(x))
    lm = Mean()
    metric = FactorizedTopK(ks=(1, 10), temperature=0.1, candidates=corpus)
    task = Retrieval(metrics=metric, loss_metrics=[lm], temperature=0.1)
    loss = task(q, corpus[idx], candidate_ids=idx)
    loss.backward()
    assert q.grad is not None and torch.isfinite(q.grad).all() and len(lm.v) == 1 and abs(lm.v[0] - loss.item()) < 1e-3
    res = task.factorized_metrics.result()
    s = (q.detach().double() @ corpus.double().T) * 10.0
    rank = (s > s[torch.arange(128), idx][:, None]).sum(1)
    assert abs(res["recall@1"] - (rank < 1).double().mean().item()) < 1e-9
    assert abs(res["recall@10"] - (rank < 10).double().mean().item()) < 1e-9
    task(q.detach(), corpus[idx], candidate_ids=idx, compute_metrics=False)       # no update
    assert metric._n == 128
    with pytest.raises(ValueError):
        task(q.detach(), corpus[idx])                                              # metrics need the candidates' corpus index
