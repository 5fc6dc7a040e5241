"""CPU tests of the oracle itself: golden id encoding (pinned by the reference), generator
known answers, analytic gradients vs finite differences, and an independent torch-CPU
restatement of the loss.  No GPU, no compute calls into the HIP library."""
import pathlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import id_encoding, synth, two_tower as tt

GOLD = pathlib.Path(__file__).resolve().parent / "golden"


# ------------------------------------------------------------------ a6 / a7: pinned by the reference
def test_id_encoding_matches_reference_golden():
    g = np.load(GOLD / "id_encoding_small.npz")
    users, items = g["user_id"].tolist(), g["parent_asin"].tolist()
    u, uv = id_encoding.encode_ids(users)
    i, iv = id_encoding.encode_ids(items)
    assert u.dtype == np.int64
    # prepare_training_data.py:113-123,209-210
    assert np.array_equal(u, g["user_idx"]) and np.array_equal(i, g["item_idx"])
    # preprocessor.py:478-491 (LabelEncoder)
    assert np.array_equal(u, g["user_id_encoded"]) and np.array_equal(i, g["item_id_encoded"])
    cats = [None if c == "<NaN>" else c for c in g["main_category"].tolist()]
    c, cv = id_encoding.encode_categories(cats)
    assert np.array_equal(c, g["category_encoded"]) and "Unknown" in cv
    # a byte-wise (UTF-8) encoder gives the same ranks, incl. non-ASCII ids
    assert any(ord(ch) > 127 for s in users for ch in s)
    assert np.array_equal(id_encoding.encode_ids_utf8(users), g["user_idx"])
    assert len(uv) == u.max() + 1 and len(iv) == i.max() + 1


# ------------------------------------------------------------------ generator
def test_splitmix_known_answers():
    # splitmix64 reference sequence for state 0: first outputs of the published generator
    # (mix(x) here = next() of a generator whose state was x)
    assert int(synth.mix(np.array([0], dtype=np.uint64))[0]) == 0xE220A8397B1DCDAF
    assert int(synth.mix(np.array([0x9E3779B97F4A7C15], dtype=np.uint64))[0]) == 0x6E789E6AA1B965F4


def test_generator_ranges_and_determinism():
    a = synth.uniform_f32(1003, synth.TID_USER_TABLE, 10000, -0.05, 0.1)
    b = synth.uniform_f32(1003, synth.TID_USER_TABLE, 10000, -0.05, 0.1)
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert a.min() >= -0.05 and a.max() < 0.05 + 1e-9 and abs(a.mean()) < 2e-3
    # windows of one stream agree with the whole stream (row-sharded regeneration)
    assert np.array_equal(synth.uniform_f32(7, 2, 100, 0.0, 1.0, start=50), synth.uniform_f32(7, 2, 150, 0.0, 1.0)[50:])
    t = synth.embedding_table(5, 1, 1000, 32)
    assert np.array_equal(synth.embedding_table(5, 1, 1000, 32, row_start=200, row_count=10), t[200:210])
    ids = synth.ids_uniform(1, 3, 50000, 1000)
    assert ids.min() >= 0 and ids.max() < 1000 and len(np.unique(ids)) == 1000
    z = synth.ids_powerlaw(1, 3, 50000, 10_000_000)
    assert z.min() >= 0 and z.max() < 10_000_000
    assert (z == 0).mean() > 0.01            # heavy head: P(id==0) = N^(-1/4) = 1.8 %
    assert not np.array_equal(synth.ids_uniform(1, 3, 100, 1000), synth.ids_uniform(1, 4, 100, 1000))


# ------------------------------------------------------------------ loss vs an independent restatement
def _torch_loss(q, c, T, w=None, p=None, ids=None, off=0):
    q, c = torch.tensor(q, dtype=torch.float64, requires_grad=True), torch.tensor(c, dtype=torch.float64, requires_grad=True)
    s = q @ c.T / T
    if p is not None:
        s = s - torch.log(torch.clamp(torch.tensor(p, dtype=torch.float64), 1e-6, 1.0))[None, :]
    nq = q.shape[0]
    lab = torch.arange(nq) + off
    if ids is not None:
        idt = torch.tensor(ids)
        dup = idt[lab][:, None] == idt[None, :]
        dup[torch.arange(nq), lab] = False
        s = s + dup.double() * tt.MIN_FLOAT
    per = F.cross_entropy(s, lab, reduction="none")
    if w is not None:
        per = per * torch.tensor(w, dtype=torch.float64)
    loss = per.sum()
    loss.backward()
    return loss.item(), q.grad.numpy(), c.grad.numpy()


@pytest.mark.parametrize("opts", [dict(), dict(w=True), dict(p=True), dict(ids=True), dict(w=True, p=True, ids=True, off=3)])
def test_retrieval_loss_and_grad_match_torch_cpu(opts):
    rng = np.random.default_rng(0)
    nq, off = 37, opts.get("off", 0)
    nc, d, T = nq + off + 5, 16, 0.1
    q, c = rng.normal(size=(nq, d)) * 0.3, rng.normal(size=(nc, d)) * 0.3
    w = rng.uniform(0.5, 2.0, nq) if opts.get("w") else None
    p = rng.uniform(1e-7, 0.3, nc) if opts.get("p") else None
    ids = rng.integers(0, 12, nc) if opts.get("ids") else None
    kw = dict(temperature=T, sample_weight=w, candidate_sampling_probability=p, candidate_ids=ids,
              remove_accidental_hits=ids is not None, diag_offset=off)
    loss, per_row, lse = tt.retrieval_loss(q, c, **kw)
    dq, dc = tt.retrieval_grad(q, c, **kw)
    tl, tdq, tdc = _torch_loss(q, c, T, w, p, ids, off)
    assert abs(loss - tl) <= 1e-9 * abs(tl)
    assert np.allclose(dq, tdq, rtol=1e-9, atol=1e-12) and np.allclose(dc, tdc, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("k", [1, 5, 40])
def test_hard_negative_mining_matches_topk_restatement(k):
    """tfrs.layers.loss.HardNegativeMining restated with torch.topk: logits + labels*MAX, top (k+1), gather, CE."""
    rng = np.random.default_rng(4)
    nq, nc, d, T = 23, 31, 8, 0.2
    q, c = rng.normal(size=(nq, d)), rng.normal(size=(nc, d))
    loss, per_row, _ = tt.retrieval_loss(q, c, temperature=T, num_hard_negatives=k)
    dq, dc = tt.retrieval_grad(q, c, temperature=T, num_hard_negatives=k)
    tq = torch.tensor(q, requires_grad=True); tc = torch.tensor(c, requires_grad=True)
    s = tq @ tc.T / T
    labels = torch.eye(nq, nc, dtype=torch.float64)
    _, idx = torch.topk(s + labels * 1e30, k=min(k + 1, nc), dim=1, sorted=False)
    sl, ll = torch.gather(s, 1, idx), torch.gather(labels, 1, idx)
    tl = -(ll * torch.log_softmax(sl, dim=1)).sum()
    tl.backward()
    assert abs(loss - tl.item()) <= 1e-9 * abs(tl.item())
    assert np.allclose(dq, tq.grad.numpy(), atol=1e-10) and np.allclose(dc, tc.grad.numpy(), atol=1e-10)


def test_accidental_hits_require_ids():
    q = np.zeros((4, 8)); c = np.zeros((4, 8))
    with pytest.raises(ValueError, match="candidate ids must be supplied"):
        tt.retrieval_loss(q, c, remove_accidental_hits=True)


# ------------------------------------------------------------------ towers + whole step: finite differences
def test_full_step_gradients_by_finite_differences():
    st = tt.synthetic_state(seed=11, n_users=50, n_items=40, emb_dim=8, tower_dims=[12, 8])
    uid = synth.ids_uniform(11, synth.TID_USER_IDS, 16, 50)
    iid = synth.ids_uniform(11, synth.TID_ITEM_IDS, 16, 40)
    r = tt.forward_backward(st, uid, iid, temperature=0.5, l2=1e-3)
    eps = 1e-6

    def total():
        return tt.forward_backward(st, uid, iid, temperature=0.5, l2=1e-3)["total"]

    for arr, grad in ((st.user_tower.weights[0], r["udw"][0]), (st.item_tower.weights[1], r["idw"][1]),
                      (st.user_tower.biases[0], r["udb"][0]), (st.item_tower.biases[1], r["idb"][1])):
        idx = tuple(np.array(arr.shape) // 2)
        old = arr[idx]
        arr[idx] = old + eps; lp = total()
        arr[idx] = old - eps; lm = total()
        arr[idx] = old
        assert abs((lp - lm) / (2 * eps) - grad[idx]) < 1e-5 * max(1.0, abs(grad[idx]))
    # embedding row gradient = sum over the positions that looked the row up
    row = int(uid[0])
    old = st.user_table[row, 3]
    st.user_table[row, 3] = old + eps; lp = total()
    st.user_table[row, 3] = old - eps; lm = total()
    st.user_table[row, 3] = old
    assert abs((lp - lm) / (2 * eps) - r["due"][uid == row][:, 3].sum()) < 1e-5


# ------------------------------------------------------------------ optimizer semantics
def test_sparse_optimizers_sum_duplicates_before_update():
    rng = np.random.default_rng(3)
    table = rng.normal(size=(10, 4)).astype(np.float32)
    ids = np.array([3, 7, 3, 3, 1], dtype=np.int64)
    g = rng.normal(size=(5, 4)).astype(np.float32)
    t1 = tt.sparse_sgd(table.copy(), ids, g, 0.1)
    gs = (g[0] + g[2]) + g[3]                      # ascending position order, f32
    assert np.array_equal(t1[3], table[3] - np.float32(0.1) * gs)
    assert np.array_equal(t1[0], table[0])
    acc = np.full_like(table, 0.1)
    t2, a2 = tt.sparse_adagrad(table.copy(), acc.copy(), ids, g, 0.1, 1e-7)
    a3 = np.float32(0.1) + gs * gs                 # g summed first: Adagrad is non-linear in g
    assert np.array_equal(a2[3], a3)
    assert np.array_equal(t2[3], table[3] - (np.float32(0.1) * gs) / np.sqrt(a3 + np.float32(1e-7)))
    wrong = np.float32(0.1) + g[0] ** 2 + g[2] ** 2 + g[3] ** 2
    assert not np.allclose(a2[3], wrong)


def test_dedup_sum_order_is_sequential_inside_a_block_and_piecewise_across():
    rng = np.random.default_rng(7)
    ids = np.concatenate([np.full(3, 5), np.full(200, 9), rng.integers(20, 40, 100)]).astype(np.int64)
    rng.shuffle(ids)
    g = rng.normal(size=(len(ids), 3)).astype(np.float32)
    uniq, s = tt.dedup_sum(ids, g)
    assert np.array_equal(uniq, np.unique(ids))
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    # id 5: three rows, all inside sorted slots [0, 64): plain sequential sum in position order
    p5 = np.flatnonzero(ids == 5)
    assert np.array_equal(s[uniq == 5][0], (g[p5[0]] + g[p5[1]]) + g[p5[2]])
    # id 9: 200 rows starting at sorted slot 3: pieces [3,64) [64,128) [128,192) [192,203)
    slots = np.flatnonzero(sid == 9)
    assert slots[0] == 3 and len(slots) == 200
    pieces = [slots[(slots // 64) == b] for b in range(4)]
    tot = None
    for pc in pieces:
        acc = g[order[pc[0]]].copy()
        for k in pc[1:]:
            acc = acc + g[order[k]]
        tot = acc if tot is None else tot + acc
    assert np.array_equal(s[uniq == 9][0], tot)
    assert np.allclose(s[uniq == 9][0], g[ids == 9].astype(np.float64).sum(0), rtol=1e-5)


def test_train_step_decreases_loss_cfg1_shape():
    st = tt.synthetic_state(seed=1001, n_users=1000, n_items=1000, emb_dim=32, tower_dims=[32], dtype=np.float64)
    uid = synth.ids_uniform(1001, synth.TID_USER_IDS, 256, 1000)
    iid = synth.ids_uniform(1001, synth.TID_ITEM_IDS, 256, 1000)
    l0 = tt.train_step(st, uid, iid, lr=0.001, optimizer="sgd")["loss"]
    for _ in range(5):
        l1 = tt.train_step(st, uid, iid, lr=0.001, optimizer="sgd")["loss"]
    assert l1 < l0
    assert abs(l0 / 256 - np.log(256)) < 0.2       # near-uniform softmax at init


def test_hash_buckets_known_answers_and_range():
    """FNV-1a-64 published test vectors (the draft's "", "a", "foobar") pin the restated hash; buckets are in range."""
    from oracle import hashing
    assert hashing.fnv1a64(b"") == 0xCBF29CE484222325
    assert hashing.fnv1a64(b"a") == 0xAF63DC4C8601EC8C
    assert hashing.fnv1a64(b"foobar") == 0x85944171F73967E8
    cats = ["All_Beauty", "Books", "Electronics", "Unknown", "Caf\u00e9"]
    b = hashing.hash_buckets(cats, 30)
    assert b.dtype == np.int64 and ((b >= 0) & (b < 30)).all()
    assert np.array_equal(b, hashing.hash_buckets(cats, 30))


def test_category_feature_gradient_is_the_item_input_gradient():
    """ie = item_row + category_row: d loss / d category_row summed over the pairs of a bucket, checked by finite
    differences on the f64 oracle; the train step moves exactly the touched buckets."""
    st = tt.synthetic_state(7, 50, 40, 8, [8], dtype=np.float64, n_category_buckets=5)
    rng = np.random.default_rng(0)
    u, i, c = rng.integers(0, 50, 16), rng.integers(0, 40, 16), rng.integers(0, 4, 16)      # bucket 4 untouched
    r = tt.forward_backward(st, u, i, temperature=0.5, category_ids=c)
    uniq, g = tt.dedup_sum(c, r["die"])
    eps = 1e-6
    for bucket, col in ((int(uniq[0]), 3), (int(uniq[-1]), 0)):
        st.cat_table[bucket, col] += eps
        lp = tt.forward_backward(st, u, i, temperature=0.5, category_ids=c)["loss"]
        st.cat_table[bucket, col] -= 2 * eps
        lm = tt.forward_backward(st, u, i, temperature=0.5, category_ids=c)["loss"]
        st.cat_table[bucket, col] += eps
        assert abs((lp - lm) / (2 * eps) - g[list(uniq).index(bucket), col]) < 1e-6
    before = st.cat_table.copy()
    tt.train_step(st, u, i, lr=0.1, optimizer="sgd", temperature=0.5, category_ids=c)
    changed = np.flatnonzero((st.cat_table != before).any(axis=1))
    assert set(changed) == set(uniq.tolist()) and 4 not in changed
