"""NumPy restatement of the two-tower retrieval training step (oracle).

TEST INFRASTRUCTURE — parity unpinned for this file (see oracle/__init__.py).

What each function follows:

* hyper-parameters / shapes: ``/root/reference/configs/data_config.yaml:54-71``
  (embedding_dim, *_tower_dims, l2_regularization, retrieval.temperature,
  retrieval.candidate_sampling = "in_batch");
* task semantics: ``tfrs.tasks.Retrieval.call`` of tensorflow-recommenders
  0.7.x (declared at ``pyproject.toml:24``, never imported by the reference):
  scores = q @ c.T; scores /= temperature; sampling-probability correction;
  accidental-hit removal; CategoricalCrossentropy(from_logits, reduction=SUM);
* optimizers: Keras 2.15 SGD / Adagrad with sparse (IndexedSlices) gradients
  de-duplicated by summation before the update.

Every function takes ``dtype``: float64 is the reference result the GPU path
is compared with; float32 mirrors the GPU's arithmetic type where bit-exact
comparisons are made (gather, SGD rows).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

MIN_FLOAT = np.finfo(np.float32).min / 100.0     # tfrs.layers.loss.MIN_FLOAT


# --------------------------------------------------------------------------- a1
def embedding_gather(table: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """E[b,:] = T[id[b],:].  TF's CPU gather raises on out-of-range ids."""
    ids = np.asarray(ids)
    if ids.size and (ids.min() < 0 or ids.max() >= table.shape[0]):
        raise IndexError("embedding id out of range")
    return table[ids]


# --------------------------------------------------------------------------- a2
def dense_fwd(x, w, b, relu: bool):
    y = x @ w + b
    return np.maximum(y, 0) if relu else y


def dense_bwd(x, w, y, dy, relu: bool, mask=None):
    """Returns dx, dw, db for y = act(x@w+b).  ``mask`` overrides the ReLU derivative (y > 0): the
    derivative is discontinuous at 0, so a test that compares gradients element-wise passes the mask
    the device actually used (pre-activations within rounding of 0 may differ in sign between f32 and f64)."""
    if relu:
        dy = dy * ((y > 0) if mask is None else mask)
    return dy @ w.T, x.T @ dy, dy.sum(axis=0)


def tower_fwd(x, weights, biases, dropout=None):
    """ReLU on all but the last layer.  ``dropout``: optional list, one (keep, scale) per hidden
    layer (oracle.synth.dropout_keep): inverted dropout on the hidden activations (Keras Dropout after each hidden
    Dense; configs/data_config.yaml:58).  Returns activations list (post-dropout)."""
    acts = [x]
    n = len(weights)
    for l, (w, b) in enumerate(zip(weights, biases)):
        y = dense_fwd(acts[-1], w, b, relu=(l < n - 1))
        if dropout is not None and l < n - 1:
            keep, scale = dropout[l]
            y = np.where(keep, y * y.dtype.type(scale), 0)
        acts.append(y)
    return acts


def tower_bwd(acts, weights, dy, masks=None, scale=1.0):
    """masks: optional list (one per hidden layer) of boolean arrays replacing (acts[l+1] > 0).
    scale: 1/(1-rate) under dropout — (acts[l+1] > 0) marks the kept active units, whose derivative is ``scale``."""
    n = len(weights)
    dws, dbs = [None] * n, [None] * n
    for l in range(n - 1, -1, -1):
        m = None if (masks is None or l >= n - 1) else masks[l]
        if l < n - 1:
            m = ((acts[l + 1] > 0) if m is None else m) * dy.dtype.type(scale)
        dy, dws[l], dbs[l] = dense_bwd(acts[l], weights[l], acts[l + 1], dy, relu=(l < n - 1), mask=m)
    return dy, dws, dbs


# ------------------------------------------------------------------------ a3+a4
def retrieval_logits(q, c, temperature=None, candidate_sampling_probability=None,
                     candidate_ids=None, remove_accidental_hits=False, diag_offset=0, num_hard_negatives=None):
    """Logits exactly as tfrs.tasks.Retrieval builds them.  Positive of query i
    is candidate ``i + diag_offset`` (diag_offset != 0 only for the sharded
    multi-GPU slab, where local queries face all-gathered candidates)."""
    s = q @ c.T
    if temperature is not None:
        s = s / temperature
    if candidate_sampling_probability is not None:
        p = np.clip(np.asarray(candidate_sampling_probability, dtype=s.dtype), 1e-6, 1.0)
        s = s - np.log(p)[None, :]
    if remove_accidental_hits:
        if candidate_ids is None:
            raise ValueError("When accidental hit removal is enabled, candidate ids must be supplied.")
        cid = np.asarray(candidate_ids)
        nq = s.shape[0]
        pos = np.arange(nq) + diag_offset
        dup = (cid[pos][:, None] == cid[None, :])
        dup[np.arange(nq), pos] = False
        s = np.where(dup, s + MIN_FLOAT, s)
    if num_hard_negatives is not None:
        # tfrs.layers.loss.HardNegativeMining: keep the positive and the k highest-scoring negatives of every row;
        # everything else leaves the softmax (here: -inf).  Negatives tied with the k-th are all kept.
        nq = s.shape[0]
        pos = np.arange(nq) + diag_offset
        neg = s.copy()
        neg[np.arange(nq), pos] = -np.inf
        if remove_accidental_hits:
            neg[dup] = -np.inf
        k = num_hard_negatives
        if k < neg.shape[1]:
            kth = np.partition(neg, neg.shape[1] - k, axis=1)[:, neg.shape[1] - k]      # k-th largest negative per row
            drop = neg < kth[:, None]
            drop[np.arange(nq), pos] = False
            s = np.where(drop, -np.inf, s)
    return s


def retrieval_loss(q, c, temperature=None, sample_weight=None,
                   candidate_sampling_probability=None, candidate_ids=None,
                   remove_accidental_hits=False, diag_offset=0, dtype=np.float64, num_hard_negatives=None):
    """Returns (loss_sum, per_row_loss, lse).  loss = sum_i w_i (lse_i - s_ii)."""
    q = np.asarray(q, dtype=dtype)
    c = np.asarray(c, dtype=dtype)
    s = retrieval_logits(q, c, temperature, candidate_sampling_probability,
                         candidate_ids, remove_accidental_hits, diag_offset, num_hard_negatives)
    nq = s.shape[0]
    m = s.max(axis=1, keepdims=True)
    lse = (m + np.log(np.exp(s - m).sum(axis=1, keepdims=True)))[:, 0]
    pos = s[np.arange(nq), np.arange(nq) + diag_offset]
    per_row = lse - pos
    if sample_weight is not None:
        per_row = per_row * np.asarray(sample_weight, dtype=dtype)
    return per_row.sum(), per_row, lse


def retrieval_grad(q, c, temperature=None, sample_weight=None,
                   candidate_sampling_probability=None, candidate_ids=None,
                   remove_accidental_hits=False, diag_offset=0, dtype=np.float64, num_hard_negatives=None):
    """d(loss_sum)/dq, d(loss_sum)/dc."""
    q = np.asarray(q, dtype=dtype)
    c = np.asarray(c, dtype=dtype)
    s = retrieval_logits(q, c, temperature, candidate_sampling_probability,
                         candidate_ids, remove_accidental_hits, diag_offset, num_hard_negatives)
    nq = s.shape[0]
    m = s.max(axis=1, keepdims=True)
    p = np.exp(s - m)
    p /= p.sum(axis=1, keepdims=True)
    p[np.arange(nq), np.arange(nq) + diag_offset] -= 1.0
    if sample_weight is not None:
        p = p * np.asarray(sample_weight, dtype=dtype)[:, None]
    if temperature is not None:
        p = p / temperature
    return p @ c, p.T @ q


def retrieval_rank_bounds(q, c, pos_index, temperature=None, candidate_sampling_probability=None, eps=1e-5):
    """(lo, hi) bounds of rank_i = #{j != pos_i : s_ij > s_i,pos_i} in f64: lo counts logits above pos + eps,
    hi those above pos - eps, so an f32 evaluation must land in [lo, hi]."""
    q = np.asarray(q, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    s = retrieval_logits(q, c, temperature, candidate_sampling_probability)
    n = s.shape[0]
    pos = s[np.arange(n), pos_index]
    s[np.arange(n), pos_index] = -np.inf
    return (s > (pos + eps)[:, None]).sum(1), (s > (pos - eps)[:, None]).sum(1)


# --------------------------------------------------------------------------- a5
PIECE = 64      # sorted slots per piece (csrc/sparse.hip kPiece)


def dedup_sum(ids, grads):
    """IndexedSlices de-duplication: rows with equal id are summed.  Order of the additions (it matters in f32, and
    the GPU kernel is bit-exact against this): stable-sort the positions by id; the run of an id occupies
    consecutive sorted slots; cut it at global multiples of PIECE slots; sum each piece sequentially in slot
    (= ascending position) order; add the pieces in order.  A run inside one 64-slot block is a plain sequential
    sum over ascending positions (what np.add.at alone would do).  Returns (unique_ids ascending, summed rows)."""
    ids = np.asarray(ids)
    n = len(ids)
    if n == 0:
        return ids[:0], grads[:0]
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    head = np.ones(n, dtype=bool)
    head[1:] = sid[1:] != sid[:-1]
    run_idx = np.cumsum(head) - 1
    piece_start = head | (np.arange(n) % PIECE == 0)
    piece_idx = np.cumsum(piece_start) - 1
    psum = np.zeros((piece_idx[-1] + 1, grads.shape[1]), dtype=grads.dtype)
    np.add.at(psum, piece_idx, grads[order])             # sequential inside a piece (0 + g == g exactly)
    out = np.zeros((run_idx[-1] + 1, grads.shape[1]), dtype=grads.dtype)
    np.add.at(out, run_idx[piece_start], psum)           # pieces in order
    return sid[head], out


def sparse_sgd(table, ids, grads, lr):
    """In place: w[u] = w[u] - fl(lr * g_sum[u])   (two roundings)."""
    uniq, g = dedup_sum(ids, grads)
    lr = table.dtype.type(lr)
    table[uniq] = table[uniq] - lr * g
    return table


def sparse_adagrad(table, accum, ids, grads, lr, eps=1e-7):
    """Keras 2.15 Adagrad, sparse path: acc += g^2; w -= lr*g / sqrt(acc + eps)."""
    uniq, g = dedup_sum(ids, grads)
    t = table.dtype.type
    a = accum[uniq] + g * g
    accum[uniq] = a
    table[uniq] = table[uniq] - (t(lr) * g) / np.sqrt(a + t(eps))
    return table, accum


def dense_sgd(w, g, lr):
    w -= w.dtype.type(lr) * g
    return w


def dense_adagrad(w, acc, g, lr, eps=1e-7):
    t = w.dtype.type
    acc += g * g
    w -= (t(lr) * g) / np.sqrt(acc + t(eps))
    return w, acc


# ------------------------------------------------------------------ whole step
@dataclass
class TowerParams:
    weights: list
    biases: list
    w_accum: list = field(default_factory=list)
    b_accum: list = field(default_factory=list)


@dataclass
class ModelState:
    user_table: np.ndarray
    item_table: np.ndarray
    user_tower: TowerParams
    item_tower: TowerParams
    user_accum: np.ndarray | None = None
    item_accum: np.ndarray | None = None
    # hashed category feature summed into the item tower's input (BASELINE configs[4]); None = no such feature
    cat_table: np.ndarray | None = None
    cat_accum: np.ndarray | None = None


def init_adagrad_state(state: ModelState, initial_accumulator_value=0.1):
    t = state.user_table.dtype.type
    state.user_accum = np.full_like(state.user_table, t(initial_accumulator_value))
    state.item_accum = np.full_like(state.item_table, t(initial_accumulator_value))
    if state.cat_table is not None:
        state.cat_accum = np.full_like(state.cat_table, t(initial_accumulator_value))
    for tw in (state.user_tower, state.item_tower):
        tw.w_accum = [np.full_like(w, t(initial_accumulator_value)) for w in tw.weights]
        tw.b_accum = [np.full_like(b, t(initial_accumulator_value)) for b in tw.biases]


def forward_backward(state: ModelState, user_ids, item_ids, temperature=0.1,
                     l2=0.0, sample_weight=None, candidate_sampling_probability=None,
                     candidate_ids=None, remove_accidental_hits=False, relu_masks=None, dropout=None,
                     category_ids=None):
    """One forward+backward.  total_loss = retrieval loss (SUM) + l2 * sum(W**2)
    over Dense kernels (Keras ``kernel_regularizer=l2``; biases unregularised).
    category_ids: bucket of every pair's hashed category; its embedding is ADDED to the item embedding
    (one add, in the state's dtype), so ``die`` is the gradient of both lookups."""
    ue = embedding_gather(state.user_table, user_ids)
    ie = embedding_gather(state.item_table, item_ids)
    if category_ids is not None:
        ie = ie + embedding_gather(state.cat_table, category_ids)
    # dropout = (user_list, item_list, scale): per-tower lists of (keep, scale) per hidden layer
    ud, idr, dscale = (None, None, 1.0) if dropout is None else dropout
    ua = tower_fwd(ue, state.user_tower.weights, state.user_tower.biases, ud)
    ia = tower_fwd(ie, state.item_tower.weights, state.item_tower.biases, idr)
    q, c = ua[-1], ia[-1]
    dt = q.dtype
    kw = dict(temperature=temperature, sample_weight=sample_weight,
              candidate_sampling_probability=candidate_sampling_probability,
              candidate_ids=candidate_ids, remove_accidental_hits=remove_accidental_hits,
              dtype=dt)
    loss, per_row, lse = retrieval_loss(q, c, **kw)
    dq, dc = retrieval_grad(q, c, **kw)
    um, im = (None, None) if relu_masks is None else relu_masks
    due, udw, udb = tower_bwd(ua, state.user_tower.weights, dq, um, dscale)
    die, idw, idb = tower_bwd(ia, state.item_tower.weights, dc, im, dscale)
    reg = dt.type(0)
    if l2:
        for tw, dws in ((state.user_tower, udw), (state.item_tower, idw)):
            for l, w in enumerate(tw.weights):
                reg = reg + dt.type(l2) * (w * w).sum()
                dws[l] = dws[l] + dt.type(2 * l2) * w
    return dict(loss=loss, reg=reg, total=loss + reg, per_row=per_row, lse=lse, q=q, c=c,
                dq=dq, dc=dc, due=due, die=die, udw=udw, udb=udb, idw=idw, idb=idb,
                user_acts=ua, item_acts=ia)


def train_step(state: ModelState, user_ids, item_ids, lr, optimizer="sgd",
               temperature=0.1, l2=0.0, eps=1e-7, **loss_kw):
    """In-place train step; returns forward_backward's dict."""
    r = forward_backward(state, user_ids, item_ids, temperature=temperature, l2=l2, **loss_kw)
    cat = loss_kw.get("category_ids")
    if optimizer == "sgd":
        sparse_sgd(state.user_table, user_ids, r["due"], lr)
        sparse_sgd(state.item_table, item_ids, r["die"], lr)
        if cat is not None:
            sparse_sgd(state.cat_table, cat, r["die"], lr)
        for tw, dws, dbs in ((state.user_tower, r["udw"], r["udb"]),
                             (state.item_tower, r["idw"], r["idb"])):
            for l in range(len(tw.weights)):
                dense_sgd(tw.weights[l], dws[l], lr)
                dense_sgd(tw.biases[l], dbs[l], lr)
    elif optimizer == "adagrad":
        sparse_adagrad(state.user_table, state.user_accum, user_ids, r["due"], lr, eps)
        sparse_adagrad(state.item_table, state.item_accum, item_ids, r["die"], lr, eps)
        if cat is not None:
            sparse_adagrad(state.cat_table, state.cat_accum, cat, r["die"], lr, eps)
        for tw, dws, dbs in ((state.user_tower, r["udw"], r["udb"]),
                             (state.item_tower, r["idw"], r["idb"])):
            for l in range(len(tw.weights)):
                dense_adagrad(tw.weights[l], tw.w_accum[l], dws[l], lr, eps)
                dense_adagrad(tw.biases[l], tw.b_accum[l], dbs[l], lr, eps)
    else:
        raise ValueError(f"unknown optimizer {optimizer!r}")
    return r


def synthetic_state(seed, n_users, n_items, emb_dim, tower_dims, dtype=np.float64,
                    optimizer="sgd", item_tower_dims=None, n_category_buckets=0) -> ModelState:
    """Deterministic init shared with the product's ``synthetic`` initialiser
    (oracle.synth tensor-id convention).  ``tower_dims`` = output dims of each
    Dense layer, e.g. [256, 128]; both towers share the shape."""
    from . import synth
    ut = synth.embedding_table(seed, synth.TID_USER_TABLE, n_users, emb_dim).astype(dtype)
    it = synth.embedding_table(seed, synth.TID_ITEM_TABLE, n_items, emb_dim).astype(dtype)
    towers = []
    for t in (0, 1):
        ws, bs, fan_in = [], [], emb_dim
        for l, fan_out in enumerate(tower_dims if (t == 0 or item_tower_dims is None) else item_tower_dims):
            ws.append(synth.dense_kernel(seed, synth.dense_tid(t, l), fan_in, fan_out).astype(dtype))
            bs.append(np.zeros(fan_out, dtype=dtype))
            fan_in = fan_out
        towers.append(TowerParams(ws, bs))
    st = ModelState(ut, it, towers[0], towers[1])
    if n_category_buckets:
        st.cat_table = synth.embedding_table(seed, synth.TID_CATEGORY_TABLE, n_category_buckets, emb_dim).astype(dtype)
    if optimizer == "adagrad":
        init_adagrad_state(st)
    return st
