"""Single-GPU train step of the two-tower retrieval model: the whole hot path on HIP kernels.

This is the explicit (no autograd) engine that ``train.py`` and ``bench.py`` drive:

    ids ──► tower fwd, the embedding lookup fused into its input tile (both layers of both towers: 1 launch)
        ──► fused scorer + softmax loss (pass 1: loss + dq, combine, pass 2: dc, slab reduction: 4 launches)
        ──► tower bwd (dx + dw + db per layer: 2 launches)
        ──► optimizer (sort of the ids, duplicate sums, sparse SGD/Adagrad of all tables, dense update: 1 launch)

All buffers are allocated once for a fixed batch size; ``step`` enqueues those 8 launches (cfg3; one more per extra tower
layer) through ONE C call (``tt_train_step_f32``) on the current stream and never synchronises (it can be captured in a
HIP graph).

Reference anchors: hyper-parameters are the ``model:`` block of
``/root/reference/configs/data_config.yaml:54-71`` (embedding_dim, *_tower_dims, l2_regularization,
training.learning_rate, retrieval.temperature, candidate_sampling "in_batch"); inputs are the int64
``user_idx`` / ``item_idx`` columns of ``prepare_training_data.py:209-210``.  The reference never
implemented the step itself (``src/training/__init__.py:1``).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

import torch

from . import _lib, ops

# tensor-id convention of the synthetic initialiser (must match oracle/synth.py, which restates it
# for the tests; the product does not import the oracle)
TID_USER_TABLE, TID_ITEM_TABLE, TID_USER_IDS, TID_ITEM_IDS = 1, 2, 3, 4
TID_CATEGORY_TABLE, TID_CATEGORY_IDS = 5, 6
TID_DENSE_BASE = 16
TID_DROPOUT_BASE = 64


@dataclass
class TwoTowerConfig:
    n_users: int
    n_items: int
    embedding_dim: int = 128                       # configs/data_config.yaml:55
    tower_dims: list = field(default_factory=lambda: [512, 256, 128])   # :56 user_tower_dims (and the item tower's if below is None)
    item_tower_dims: list | None = None            # :57 item_tower_dims; None = same as tower_dims
    temperature: float = 0.1                       # :70
    l2_regularization: float = 1e-6                # :59
    learning_rate: float = 0.001                   # :63
    optimizer: str = "sgd"                         # unspecified by the reference; north_star: SGD / Adagrad
    adagrad_initial_accumulator: float = 0.1       # Keras 2.15 default
    adagrad_epsilon: float = 1e-7                  # Keras 2.15 default
    batch_size: int = 1024                         # :62
    dropout_rate: float = 0.0                      # :58 is 0.1; parity/bench runs use 0
    # BASELINE configs[4] "30 categories as hash features": a [n_category_buckets, embedding_dim] table whose row
    # (bucket of the pair's hashed category) is ADDED to the item embedding before the item tower.  0 = no such feature.
    n_category_buckets: int = 0
    # matrix products of the scorer + softmax loss: "f32" (exact f32 products on the f32-input MFMA: the parity-safe default
    # and the headline) or "bf16x3" (f32-emulated through a three-way bf16 split on the bf16 MFMA; scorer dim 128 / 256)
    scorer_precision: str = "f32"

    @property
    def user_dims(self) -> list:
        return list(self.tower_dims)

    @property
    def item_dims(self) -> list:
        return list(self.tower_dims if self.item_tower_dims is None else self.item_tower_dims)

    @property
    def symmetric(self) -> bool:
        return self.user_dims == self.item_dims

    def validate(self):
        if self.optimizer not in ("sgd", "adagrad"):
            raise ValueError(f"optimizer must be 'sgd' or 'adagrad', got {self.optimizer!r}")
        if self.embedding_dim % 4 or any(d % 4 for d in self.user_dims + self.item_dims):
            raise ValueError("embedding_dim and tower dims must be multiples of 4")
        if self.user_dims[-1] != self.item_dims[-1]:
            raise ValueError("both towers must end in the same (scorer) dimension")
        if self.tower_dims[-1] not in (32, 64, 128, 256):
            raise ValueError("the last tower dim (scorer dim) must be one of 32, 64, 128, 256")
        if not 0.0 <= self.dropout_rate < 1.0:
            raise ValueError("dropout_rate must be in [0, 1)")
        if self.temperature <= 0:
            raise ValueError("temperature must be positive")
        if self.n_category_buckets < 0:
            raise ValueError("n_category_buckets must be >= 0")
        if self.scorer_precision not in ops.SCORER_PRECISIONS:
            raise ValueError(f"scorer_precision must be one of {ops.SCORER_PRECISIONS}")
        if self.scorer_precision == "bf16x3" and self.tower_dims[-1] not in (128, 256):
            raise ValueError("scorer_precision='bf16x3' needs a scorer dim (last tower dim) of 128 or 256")


class Tower:
    """Dense stack: ReLU on all but the last layer (Keras Dense)."""

    def __init__(self, cfg: TwoTowerConfig, tower_dims: list, flat: torch.Tensor, flat_acc, offset: int, dev):
        self.dims = [cfg.embedding_dim] + list(tower_dims)
        self.n_layers = len(tower_dims)
        b = cfg.batch_size
        self.w, self.b, self.w_acc, self.b_acc = [], [], [], []
        for l in range(self.n_layers):
            k, n = self.dims[l], self.dims[l + 1]
            self.w.append(flat[offset:offset + k * n].view(k, n))
            self.w_acc.append(None if flat_acc is None else flat_acc[offset:offset + k * n].view(k, n))
            offset += k * n
            self.b.append(flat[offset:offset + n])
            self.b_acc.append(None if flat_acc is None else flat_acc[offset:offset + n])
            offset += n
        self.end_offset = offset
        ns = ops.dense_bwd_num_slabs(b)
        self.n_slabs = ns
        self.acts = [torch.empty(b, self.dims[0], device=dev)] + \
                    [torch.empty(b, self.dims[l + 1], device=dev) for l in range(self.n_layers)]
        # dz[l]: gradient w.r.t. the pre-activation of layer l; dz[n_layers-1] is the scorer's dq/dc
        self.dz = [torch.empty(b, self.dims[l + 1], device=dev) for l in range(self.n_layers)]
        self.demb = torch.empty(b, self.dims[0], device=dev)
        # sign bits of the hidden (ReLU) activations, written by the forward GEMM's epilogue and read by the next layer's dx
        # epilogue as its mask: [b, n/32] words instead of re-reading acts[l] (indexed like acts; None where n % 32 != 0)
        self.bits = [None] + [ops.relu_bits_like(b, self.dims[l + 1], dev) if (l < self.n_layers - 1 and self.dims[l + 1] % 32 == 0)
                              else None for l in range(self.n_layers)]
        self.dw_slabs = [torch.empty(ns, self.dims[l], self.dims[l + 1], device=dev) for l in range(self.n_layers)]
        self.db_slabs = [torch.empty(ns, self.dims[l + 1], device=dev) for l in range(self.n_layers)]

    @staticmethod
    def param_count(cfg: TwoTowerConfig, tower_dims: list) -> int:
        dims = [cfg.embedding_dim] + list(tower_dims)
        return sum(dims[l] * dims[l + 1] + dims[l + 1] for l in range(len(tower_dims)))

    def forward(self, dropout=None, lookup=None):
        """dropout = (rate, seed, tower_index, first_global_row) in training; None = inference (no dropout).
        Inverted dropout follows every hidden (ReLU) layer, fused in the GEMM epilogue.
        lookup (ops.make_lookup): layer 0 reads its input rows from the embedding table (acts[0] is not used)."""
        for l in range(self.n_layers):
            hidden = l < self.n_layers - 1
            d = None
            if dropout is not None and hidden and dropout[0] > 0.0:
                rate, seed, tower, row0 = dropout
                d = (rate, seed, TID_DROPOUT_BASE + 2 * l + tower, row0 * self.dims[l + 1])
            ops.dense_fwd(self.acts[l], self.w[l], self.b[l], relu=hidden, out=self.acts[l + 1], dropout=d,
                          lookup=lookup if l == 0 else None, relu_bits=self.bits[l + 1])
        return self.acts[-1]

    def backward(self, dropout_rate: float = 0.0, dx: bool = True, dw: bool = True, lookup=None):
        """Consumes dz[-1]; leaves demb (dx) and the dw/db slabs (dw).  backward(dx=True, dw=False) followed by
        backward(dx=False, dw=True) is the same computation with every dx first."""
        scale = 1.0
        if dropout_rate > 0.0:      # the same f32 arithmetic as the forward kernel's launcher: 1.0f / (1.0f - rate)
            one = torch.ones((), dtype=torch.float32)
            scale = (one / (one - torch.tensor(dropout_rate, dtype=torch.float32))).item()
        for l in range(self.n_layers - 1, -1, -1):
            dxo = (self.dz[l - 1] if l > 0 else self.demb) if dx else None
            bits = self.bits[l] if (l > 0 and dx) else None
            mask_src = self.acts[l] if (l > 0 and dx and bits is None) else None   # acts[l] = (dropped-out) ReLU output of layer l-1
            ops.dense_bwd(self.acts[l], self.w[l], self.dz[l], dxo, mask_src, self.dw_slabs[l] if dw else None,
                          self.db_slabs[l] if dw else None, dx_scale=scale if l > 0 else 1.0,
                          lookup=lookup if l == 0 else None, dx_relu_bits=bits)

    def segments(self, l2: float, grad_flat=None, grad_offset: int = 0):
        segs = []
        off = grad_offset
        for l in range(self.n_layers):
            for p, acc, slabs, reg in ((self.w[l], self.w_acc[l], self.dw_slabs[l], l2),
                                       (self.b[l], self.b_acc[l], self.db_slabs[l], 0.0)):
                gout = None if grad_flat is None else grad_flat[off:off + p.numel()]
                segs.append(ops.make_dense_seg(p, acc, slabs, self.n_slabs, reg, gout))
                off += p.numel()
        return segs


def towers_forward(ut: "Tower", it: "Tower", dropout=None, lookups=None):
    """Both towers layer by layer, one launch per layer (the towers have identical shapes).
    dropout = (rate, seed, first_global_row) in training, None at inference.
    lookups = (user lookup, item lookup): layer 0 gathers its input rows from the embedding tables itself."""
    L = ut.n_layers
    # the last two layers (ReLU hidden + linear output) of both towers in ONE launch when their shapes allow it: the hidden tile
    # never leaves the CU between them (csrc/tower.hip); two-layer towers are that launch alone (with the lookup inside), deeper
    # ones - the reference's [512, 256, 128] - run the layers below one by one first
    fused = L >= 2 and ops.tower_fwd2_supported(ut.acts[1].shape[0], ut.dims[L - 2], ut.dims[L - 1], ut.dims[L])
    for l in range(L - 2 if fused else L):
        hidden = l < L - 1
        d = None
        if dropout is not None and hidden and dropout[0] > 0.0:
            rate, seed, row0 = dropout
            d = (rate, seed, (TID_DROPOUT_BASE + 2 * l, TID_DROPOUT_BASE + 2 * l + 1), row0 * ut.dims[l + 1])
        ops.dense_fwd2((ut.acts[l], it.acts[l]), (ut.w[l], it.w[l]), (ut.b[l], it.b[l]), (ut.acts[l + 1], it.acts[l + 1]),
                       relu=hidden, dropout=d, lookups=lookups if l == 0 else None, relu_bits=(ut.bits[l + 1], it.bits[l + 1]))
    if fused:
        a = L - 2
        d = None
        if dropout is not None and dropout[0] > 0.0:
            rate, seed, row0 = dropout
            d = (rate, seed, (TID_DROPOUT_BASE + 2 * a, TID_DROPOUT_BASE + 2 * a + 1), row0 * ut.dims[a + 1])
        ops.tower_fwd2((ut.acts[a], it.acts[a]), (ut.w[a], it.w[a]), (ut.b[a], it.b[a]), (ut.acts[a + 1], it.acts[a + 1]),
                       (ut.bits[a + 1], it.bits[a + 1]), (ut.w[a + 1], it.w[a + 1]), (ut.b[a + 1], it.b[a + 1]),
                       (ut.acts[a + 2], it.acts[a + 2]), dropout=d, lookups=lookups if a == 0 else None)
    return ut.acts[-1], it.acts[-1]


def towers_backward(ut: "Tower", it: "Tower", dropout_rate: float = 0.0, on_embedding_grads=None, lookups=None, bwd2_ws=None):
    """Backward of both towers, one launch per layer (dx and dw+db tiles of both towers side by side).
    With ``on_embedding_grads`` every dx is computed first, the callback runs as soon as demb is complete (the
    sharded trainer starts the gradient exchange there) and the dw+db launches follow, beside the transfer.
    ``bwd2_ws`` (ops.tower_bwd2_workspace): layers 1 and 0 in ONE launch where the shape allows (tt_tower_bwd2_batched_f32)."""
    scale = 1.0
    if dropout_rate > 0.0:
        one = torch.ones((), dtype=torch.float32)
        scale = (one / (one - torch.tensor(dropout_rate, dtype=torch.float32))).item()
    none2 = (None, None)

    def layer_args(l, dx: bool, dw: bool):
        dxs = ((ut.dz[l - 1], it.dz[l - 1]) if l > 0 else (ut.demb, it.demb)) if dx else none2
        bits = (ut.bits[l], it.bits[l]) if (l > 0 and dx and ut.bits[l] is not None) else none2
        masks = (ut.acts[l], it.acts[l]) if (l > 0 and dx and bits[0] is None) else none2
        return dict(xs=(ut.acts[l], it.acts[l]), ws=(ut.w[l], it.w[l]), dzs=(ut.dz[l], it.dz[l]), dxs=dxs, dx_relu_srcs=masks,
                    dw_slabs=(ut.dw_slabs[l], it.dw_slabs[l]) if dw else none2, db_slabs=(ut.db_slabs[l], it.db_slabs[l]) if dw else none2,
                    lookups=lookups if l == 0 else None, dx_relu_bits=bits)

    def layer(l, dx: bool, dw: bool):
        a = layer_args(l, dx, dw)
        ops.dense_bwd2(a["xs"], a["ws"], a["dzs"], a["dxs"], a["dx_relu_srcs"], a["dw_slabs"], a["db_slabs"],
                       dx_scale=scale if l > 0 else 1.0, lookups=a["lookups"], dx_relu_bits=a["dx_relu_bits"])

    if on_embedding_grads is None:
        last = -1
        if (bwd2_ws is not None and ut.n_layers >= 2
                and ops.tower_bwd2_supported(ut.dz[0].shape[0], ut.dims[0], ut.dims[1], ut.dims[2])):
            last = 1
        for l in range(ut.n_layers - 1, last, -1):
            layer(l, True, True)
        if last == 1:
            ops.tower_bwd2(layer_args(1, True, True), layer_args(0, True, True), bwd2_ws, dx_scale_upper=scale, dx_scale_lower=1.0)
        return
    for l in range(ut.n_layers - 1, -1, -1):
        layer(l, True, False)
    on_embedding_grads()
    for l in range(ut.n_layers - 1, -1, -1):
        layer(l, False, True)


class TwoTowerTrainer:
    def __init__(self, cfg: TwoTowerConfig, device="cuda:0", seed: int | None = None):
        cfg.validate()
        self.cfg = cfg
        self.dev = dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("TwoTowerTrainer needs a CUDA/HIP device: there is no CPU fallback")
        b, d = cfg.batch_size, cfg.embedding_dim
        adagrad = cfg.optimizer == "adagrad"
        self.user_table = torch.empty(cfg.n_users, d, device=dev)
        self.item_table = torch.empty(cfg.n_items, d, device=dev)
        self.user_accum = torch.full_like(self.user_table, cfg.adagrad_initial_accumulator) if adagrad else None
        self.item_accum = torch.full_like(self.item_table, cfg.adagrad_initial_accumulator) if adagrad else None
        n_user, n_item = Tower.param_count(cfg, cfg.user_dims), Tower.param_count(cfg, cfg.item_dims)
        self.dense_flat = torch.zeros(n_user + n_item, device=dev)
        self.dense_accum = torch.full_like(self.dense_flat, cfg.adagrad_initial_accumulator) if adagrad else None
        self.dense_grad = torch.empty_like(self.dense_flat)        # summed gradients (multi-GPU all-reduce bucket)
        self.user_tower = Tower(cfg, cfg.user_dims, self.dense_flat, self.dense_accum, 0, dev)
        self.item_tower = Tower(cfg, cfg.item_dims, self.dense_flat, self.dense_accum, n_user, dev)
        sd = cfg.tower_dims[-1]
        self.ws = torch.empty(ops.retrieval_workspace_bytes(b, b, sd), dtype=torch.uint8, device=dev)
        self.lse = torch.empty(b, device=dev)
        self.per_row = torch.empty(b, device=dev)
        self.loss = torch.empty(1, device=dev)
        self.oob = torch.zeros(1, dtype=torch.int32, device=dev)
        self.user_plan = ops.SparsePlan(b, dev)
        self.item_plan = ops.SparsePlan(b, dev)
        self.cat_table = self.cat_accum = self.cat_plan = None
        if cfg.n_category_buckets:
            self.cat_table = torch.empty(cfg.n_category_buckets, d, device=dev)
            self.cat_accum = torch.full_like(self.cat_table, cfg.adagrad_initial_accumulator) if adagrad else None
            self.cat_plan = ops.SparsePlan(b, dev)
        # high priority = a hardware queue of its own (ROCm pools queues per priority): the sort plans always run BESIDE
        # the main stream's kernels, whatever other streams the process has created
        self._side = torch.cuda.Stream(device=dev, priority=-1)
        # where the sort plan runs: in front of the forward pass on the main stream (default since the partitioned sort:
        # ~6 us of kernel), or beside it on the side stream (TT_PLAN_STREAM=side: two cross-stream waits per step, which
        # cost more than the kernel: cfg3 step 0.687-0.689 ms against 0.677-0.681 ms; DESIGN.md section 4, K2 plan)
        self.plan_on_side_stream = os.environ.get("TT_PLAN_STREAM", "main") == "side"
        self.step_index = 0                      # counter of the dropout stream (global batch row = step*batch + r)
        # K1 inside the first tower layer's GEMMs (0: gather2 launch + acts[0]).  Every 64-column tile of the first layer reads
        # the embedding rows through the ids again: up to 256 columns that costs less than materialising them (cfg3: 13.6 us
        # of gather against +3.6 us in the GEMMs; cfg4 1.875 vs 1.881 ms), at 512 columns it costs more (cfg5: the layer-0
        # launches 99 us longer, the gather 55 us; step 15.28 vs 15.24 ms) - r03 A/B
        # (r04: at small batches the gather launch's fixed cost decides - the reference's own config, [512, 256, 128] at batch 1024:
        # 0.1631-0.1638 ms fused against 0.1644-0.1654 with the gather launch)
        self.fuse_lookup = os.environ.get("TT_FUSE_LOOKUP", "1" if (cfg.tower_dims[0] < 512 or cfg.batch_size <= 2048) else "0") != "0"
        self.fuse_sort = os.environ.get("TT_FUSE_SORT", "1") != "0"   # the optimizer launch sorts the ids itself (no plan launch)
        self.fuse_optimizer = True               # sparse + dense optimizer in one launch (False: dense_update, sparse_update2 [, cat])
        # the whole step behind ONE C call (tt_train_step_f32: the same launches - eight at cfg3 -, enqueued in C - one FFI crossing per step
        # instead of nine; cfg1 is host-bound).  TT_COMPOSITE_STEP=0: the Python sequence of the separate entry points.
        self.use_composite = os.environ.get("TT_COMPOSITE_STEP", "1") != "0"
        self._cstep = None
        self._id_bucket_ws = None
        # layers 1 and 0 of the backward pass in ONE launch (tt_tower_bwd2_batched_f32: the lower layer's tiles wait inside the
        # launch for the rows of dz they read); TT_BWD2=0: one launch per layer
        self.bwd2_ws = None
        if (os.environ.get("TT_BWD2", "0") != "0" and cfg.symmetric and len(cfg.tower_dims) >= 2
                and ops.tower_bwd2_supported(b, d, cfg.tower_dims[0], cfg.tower_dims[1])):
            self.bwd2_ws = ops.tower_bwd2_workspace(b, dev)
        self.flag_poll_every = 50                # steps between asynchronous polls of the out-of-range flag (0 = never)
        self._oob_host = self._oob_event = None
        self._oob_step = -1
        # r04: the one-launch optimizer gives every row range of a table to ONE workgroup.  Ids uniform over the rows put
        # batch / groups ~ 64-270 ids into each; a vocabulary in order of frequency (what StringLookup.adapt builds) puts a third
        # of a power-law batch into the first range, and that workgroup is the launch (cfg3, ids ~ rows * u^4: 12 -> 169 us, step
        # 0.560 -> 0.746 ms).  Every flag_poll_every steps a one-workgroup-per-table probe (tt_id_range_load) counts the batch's
        # ids per range; its result is read from pinned memory when it has landed, never waited for, and while some range holds
        # more than skew_limit ids the steps take the plan launch + tt_optimizer_step_f32 instead (the sorted list is spread
        # over all CUs whatever the ids: 38 us at cfg3), and return to the one launch when the load falls below 3/4 of the limit.
        # 384: the overloaded workgroup costs ~0.066 us per id at dim 128, the plan path ~25 us more than the balanced launch
        # (the reference's own config, batch 1024, power-law ids, largest range 498-541: 0.2053 ms without the probe, 0.1834 at
        # limit 512 - flapping -, 0.1694 at 384 and 256; cfg1's 177 ids must NOT switch: its Python sequence of launches is
        # host-bound, 0.060 -> 0.090 ms).  TT_SKEW_LIMIT=0: never switch.
        self.skew_limit = int(os.environ.get("TT_SKEW_LIMIT", "384"))
        self.range_load = 0                      # largest row-range load the last finished probe saw
        self._skew_state = False
        self._skew_dev = self._skew_host = self._skew_event = None
        self.dropout_seed = 0 if seed is None else seed
        self._segs = self.user_tower.segments(cfg.l2_regularization) + self.item_tower.segments(cfg.l2_regularization)
        if seed is not None:
            self.init_synthetic(seed)

    # ------------------------------------------------------------------ init
    def init_synthetic(self, seed: int):
        """Keras defaults (Embedding U(-0.05,0.05), Dense Glorot-uniform, zero bias) from the counter-based
        generator: identical, bit for bit, to oracle.two_tower.synthetic_state(seed, ...)."""
        ops.fill_uniform_(self.user_table, seed, TID_USER_TABLE, -0.05, 0.1)
        ops.fill_uniform_(self.item_table, seed, TID_ITEM_TABLE, -0.05, 0.1)
        if self.cat_table is not None:
            ops.fill_uniform_(self.cat_table, seed, TID_CATEGORY_TABLE, -0.05, 0.1)
        self.dense_flat.zero_()
        for t, tower in enumerate((self.user_tower, self.item_tower)):
            for l, w in enumerate(tower.w):
                lim = torch.tensor(math.sqrt(6.0 / (w.shape[0] + w.shape[1])), dtype=torch.float64).to(torch.float32)
                lim32 = lim.item()
                scale32 = (lim + lim).item()
                ops.fill_uniform_(w, seed, TID_DENSE_BASE + 2 * l + t, -lim32, scale32)
        if self.cfg.optimizer == "adagrad":
            for a in (self.user_accum, self.item_accum, self.dense_accum, self.cat_accum):
                if a is not None:
                    a.fill_(self.cfg.adagrad_initial_accumulator)

    def synthetic_batch(self, seed: int, step: int, variant: str = "U", out=None):
        b = self.cfg.batch_size
        if out is None:
            out = (torch.empty(b, dtype=torch.int64, device=self.dev), torch.empty(b, dtype=torch.int64, device=self.dev))
        ops.fill_ids_(out[0], seed, TID_USER_IDS, self.cfg.n_users, variant, start=step * b)
        ops.fill_ids_(out[1], seed, TID_ITEM_IDS, self.cfg.n_items, variant, start=step * b)
        return out

    def synthetic_categories(self, seed: int, step: int, variant: str = "Z", out=None):
        """Category bucket of every pair of synthetic step ``step`` (power-law by default: a few big categories)."""
        b = self.cfg.batch_size
        if out is None:
            out = torch.empty(b, dtype=torch.int64, device=self.dev)
        ops.fill_ids_(out, seed, TID_CATEGORY_IDS, self.cfg.n_category_buckets, variant, start=step * b)
        return out

    def _check_categories(self, category_ids):
        if (category_ids is None) != (self.cat_table is None):
            raise ValueError("category_ids must be given exactly when cfg.n_category_buckets > 0")
        if category_ids is not None and category_ids.numel() != self.cfg.batch_size:
            raise ValueError(f"category_ids must have {self.cfg.batch_size} entries")

    def _check_batch(self, *id_tensors):
        """The tower buffers hold exactly cfg.batch_size rows: more ids would write past them, fewer would leave stale
        rows that the scorer still reads."""
        b = self.cfg.batch_size
        for t in id_tensors:
            if t is not None and t.numel() != b:
                raise ValueError(f"batch must have exactly {b} entries (got {t.numel()}): the kernels' buffers are sized "
                                 "for cfg.batch_size; pad or drop a ragged last batch")

    def _lookups(self, user_ids, item_ids, category_ids):
        """K1 fused into the towers' first layer (fwd GEMM and dW GEMM read the table rows themselves): the
        [batch, dim] tower inputs are never written to HBM.  None when the batch is too long for the fused form."""
        self._check_batch(user_ids, item_ids, category_ids)
        if not self.fuse_lookup or self.cfg.batch_size > ops.MAX_FUSED_LOOKUP_ROWS:
            return None
        return (ops.make_lookup(self.user_table, user_ids, oob_flag=self.oob),
                ops.make_lookup(self.item_table, item_ids, self.cat_table, category_ids, self.oob))

    def _item_inputs(self, user_ids, item_ids, category_ids):
        """K1 as its own launch (fuse_lookup = False): both towers' input rows; the hashed category's row is summed
        into the item tower's input."""
        self._check_batch(user_ids, item_ids, category_ids)
        ut, it = self.user_tower, self.item_tower
        ops.embedding_gather2(self.user_table, user_ids, ut.acts[0], self.item_table, item_ids, it.acts[0], self.oob)
        if category_ids is not None:
            ops.embedding_gather_add_(it.acts[0], self.cat_table, category_ids, self.oob)

    # ------------------------------------------------------------------ the hot path
    def forward_backward(self, user_ids: torch.Tensor, item_ids: torch.Tensor, sample_weight=None,
                         candidate_sampling_probability=None, candidate_ids=None, category_ids=None):
        cfg, ut, it = self.cfg, self.user_tower, self.item_tower
        self._check_categories(category_ids)
        lks = self._lookups(user_ids, item_ids, category_ids)
        if lks is None:
            self._item_inputs(user_ids, item_ids, category_ids)
            lks = (None, None)
        row0 = self.step_index * cfg.batch_size
        if cfg.symmetric:        # same shapes: every layer of both towers in one launch
            q, c = towers_forward(ut, it, (cfg.dropout_rate, self.dropout_seed, row0), lookups=lks if lks[0] else None)
        else:
            q = ut.forward((cfg.dropout_rate, self.dropout_seed, 0, row0), lookup=lks[0])
            c = it.forward((cfg.dropout_rate, self.dropout_seed, 1, row0), lookup=lks[1])
        kw = dict(sample_weight=sample_weight, cand_prob=candidate_sampling_probability, cand_ids=candidate_ids)
        # loss + dq + dc in two fused passes over the logits (probabilities never stored; f32: raw dot products kept in self.ws)
        ops.retrieval_fwd_bwd(q, c, 1.0 / cfg.temperature, self.ws, self.lse, self.per_row, self.loss,
                              ut.dz[-1], it.dz[-1], precision=cfg.scorer_precision, **kw)
        if cfg.symmetric:
            towers_backward(ut, it, cfg.dropout_rate, lookups=lks if lks[0] else None, bwd2_ws=self.bwd2_ws)
        else:
            ut.backward(cfg.dropout_rate, lookup=lks[0])
            it.backward(cfg.dropout_rate, lookup=lks[1])
        self.step_index += 1
        return self.loss

    def apply_gradients(self, step_ids=None):
        """``step_ids`` = [user ids, item ids (, category ids)]: the optimizer launch sorts them itself (no plan launch ran)."""
        cfg = self.cfg
        if step_ids is not None:     # sort + sparse update of every table + dense update: ONE launch, straight from the raw ids
            tables = [(self.user_table, self.user_accum, self.user_tower.demb, step_ids[0], self.user_plan),
                      (self.item_table, self.item_accum, self.item_tower.demb, step_ids[1], self.item_plan)]
            if self.cat_table is not None:
                tables.append((self.cat_table, self.cat_accum, self.item_tower.demb, step_ids[2], self.cat_plan))
            ops.optimizer_step_ids_(cfg.optimizer, tables, self._segs, cfg.learning_rate, cfg.adagrad_epsilon)
            return
        if self.fuse_optimizer:      # sparse update of every table + dense update of every tower segment: one launch
            tables = [(self.user_table, self.user_accum, self.user_tower.demb, self.user_plan),
                      (self.item_table, self.item_accum, self.item_tower.demb, self.item_plan)]
            if self.cat_table is not None:   # the category row's gradient is the item-tower input gradient itself
                tables.append((self.cat_table, self.cat_accum, self.item_tower.demb, self.cat_plan))
            ops.optimizer_step_(cfg.optimizer, tables, self._segs, cfg.learning_rate, cfg.adagrad_epsilon)
            return
        ops.dense_update_(self._segs, cfg.optimizer, cfg.learning_rate, cfg.adagrad_epsilon)
        ops.sparse_update2_(cfg.optimizer, self.user_table, self.user_accum, self.user_tower.demb, self.user_plan,
                            self.item_table, self.item_accum, self.item_tower.demb, self.item_plan,
                            cfg.learning_rate, cfg.adagrad_epsilon)
        if self.cat_table is not None:       # the category row's gradient is the item-tower input gradient itself
            if cfg.optimizer == "sgd":
                ops.sparse_sgd_(self.cat_table, self.item_tower.demb, self.cat_plan, cfg.learning_rate)
            else:
                ops.sparse_adagrad_(self.cat_table, self.cat_accum, self.item_tower.demb, self.cat_plan, cfg.learning_rate,
                                    cfg.adagrad_epsilon)

    def step(self, user_ids: torch.Tensor, item_ids: torch.Tensor, **loss_kw) -> torch.Tensor:
        """One train step; returns the (device, unsynchronised) retrieval loss (SUM over the batch)."""
        self._check_batch(user_ids, item_ids, loss_kw.get("category_ids"), loss_kw.get("sample_weight"),
                          loss_kw.get("candidate_sampling_probability"), loss_kw.get("candidate_ids"))
        # (never inside a graph capture: the poll queries an event recorded outside it and would bake a D2H copy into every replay)
        if self.flag_poll_every and self.step_index % self.flag_poll_every == 0 and not torch.cuda.is_current_stream_capturing():
            self.poll_ids()
        # the sort plans depend on the ids only: one launch for all tables, in front of the forward pass (or beside it)
        main = torch.cuda.current_stream()
        plans, ids, rows = [self.user_plan, self.item_plan], [user_ids, item_ids], [self.cfg.n_users, self.cfg.n_items]
        if loss_kw.get("category_ids") is not None and self.cat_plan is not None:
            plans.append(self.cat_plan); ids.append(loss_kw["category_ids"]); rows.append(self.cfg.n_category_buckets)
        # fuse_sort: no plan launch at all - the optimizer launch's workgroups sort the ids of their own row range in LDS
        # and update exactly those rows (tt_optimizer_step_ids_f32; lists up to 65536 ids) - unless the batches are skewed
        shape_ok = (self.fuse_sort and self.fuse_optimizer and user_ids.numel() <= ops.optimizer_ids_max_ids()
                    and (self.cat_table is None) == (len(ids) == 2))
        if shape_ok:
            self._poll_skew(ids, rows)
        fused_sort = shape_ok and not self._skewed()
        if fused_sort:
            if (self.use_composite and self.fuse_lookup and self.cfg.symmetric
                    and self.cfg.batch_size <= ops.MAX_FUSED_LOOKUP_ROWS):
                return self._step_composite(ids, **loss_kw)
            loss = self.forward_backward(user_ids, item_ids, **loss_kw)
            self.apply_gradients(step_ids=ids)
            return loss
        if self.plan_on_side_stream:
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):    # one launch for all tables (csrc/sort.hip)
                ops.sparse_plan_batched(plans, ids, rows)
        else:
            ops.sparse_plan_batched(plans, ids, rows)
        loss = self.forward_backward(user_ids, item_ids, **loss_kw)
        if self.plan_on_side_stream:
            main.wait_stream(self._side)
        self.apply_gradients()
        return loss

    def _build_composite(self):
        """The step's description for tt_train_step_f32 (include/twotower_hip.h): every buffer of the step is allocated once,
        so the struct is filled once; step() rewrites the id pointers, the dropout row counter and the optional inputs."""
        cfg, ut, it = self.cfg, self.user_tower, self.item_tower
        st = _lib.TrainStep()
        st.batch, st.n_layers = cfg.batch_size, ut.n_layers
        if ut.n_layers > _lib.TT_MAX_TOWER_LAYERS or len(self._segs) > _lib.TT_MAX_DENSE_SEGS:
            return None
        for l, d in enumerate(ut.dims):
            st.dims[l] = d
        p = ops._p
        for l in range(ut.n_layers):
            for i, tw in enumerate((ut, it)):
                f, b = st.fwd[l][i], st.bwd[l][i]
                f.x = None if l == 0 else p(tw.acts[l])
                f.w, f.b, f.y = p(tw.w[l]), p(tw.b[l]), p(tw.acts[l + 1])
                f.dropout_tensor_id = TID_DROPOUT_BASE + 2 * l + i
                f.relu_bits = p(tw.bits[l + 1])
                b.x = None if l == 0 else p(tw.acts[l])
                b.w, b.dz = p(tw.w[l]), p(tw.dz[l])
                b.dx = p(tw.dz[l - 1] if l > 0 else tw.demb)
                b.dx_relu_bits = p(tw.bits[l]) if l > 0 else None
                b.dx_relu_src = p(tw.acts[l]) if (l > 0 and tw.bits[l] is None) else None
                b.dw_slabs, b.db_slabs = p(tw.dw_slabs[l]), p(tw.db_slabs[l])
        for i, (table, rows2) in enumerate(((self.user_table, None), (self.item_table, self.cat_table))):
            for lk in (st.fwd[0][i].lookup, st.bwd[0][i].lookup):
                lk.table, lk.table_rows, lk.oob_flag = p(table), table.shape[0], p(self.oob)
                if rows2 is not None:
                    lk.table2, lk.table2_rows = p(rows2), rows2.shape[0]
        st.dropout_rate, st.dropout_seed = cfg.dropout_rate, self.dropout_seed
        st.scorer_precision = ops.SCORER_PRECISIONS.index(cfg.scorer_precision)
        st.inv_temperature = 1.0 / cfg.temperature
        st.retrieval_ws, st.retrieval_ws_bytes = p(self.ws), self.ws.numel()
        st.lse, st.per_row, st.loss = p(self.lse), p(self.per_row), p(self.loss)
        st.opt = ops._OPT[cfg.optimizer]
        tabs = [(self.user_table, self.user_accum, ut.demb, self.user_plan), (self.item_table, self.item_accum, it.demb, self.item_plan)]
        if self.cat_table is not None:          # the category row's gradient is the item-tower input gradient itself
            tabs.append((self.cat_table, self.cat_accum, it.demb, self.cat_plan))
        st.n_tables = len(tabs)
        for i, (table, accum, grads, plan) in enumerate(tabs):
            t = st.tables[i]
            t.table, t.accum, t.rows, t.grads = p(table), p(accum), table.shape[0], p(grads)
            t.apply_ws = p(plan.apply_ws(cfg.embedding_dim))
        st.n_segs = len(self._segs)
        for i, seg in enumerate(self._segs):
            st.segs[i] = seg
        st.lr, st.eps = cfg.learning_rate, cfg.adagrad_epsilon
        # r04: the forward lookup hands the optimizer launch its row-range id lists (tt_id_buckets, ABI v9): a zeroed workspace
        # is all the caller supplies, tt_train_step_f32 cuts the ranges and stamps a generation per step
        if self._id_bucket_ws is None:
            per = int(_lib.load().tt_id_buckets_workspace_bytes())
            self._id_bucket_ws = torch.zeros(2 * per, dtype=torch.uint8, device=self.dev)
        st.id_bucket_ws, st.id_bucket_ws_bytes = p(self._id_bucket_ws), self._id_bucket_ws.numel()
        return st

    def _step_composite(self, ids, sample_weight=None, candidate_sampling_probability=None, candidate_ids=None,
                        category_ids=None) -> torch.Tensor:
        """forward_backward + apply_gradients as ONE call into the library (same launches, same order, same results)."""
        self._check_categories(category_ids)
        if self._cstep is None:
            self._cstep = self._build_composite()
            if self._cstep is None:                  # more layers / segments than the struct holds: the Python sequence
                self.use_composite = False
                loss = self.forward_backward(ids[0], ids[1], sample_weight=sample_weight, category_ids=category_ids,
                                             candidate_sampling_probability=candidate_sampling_probability, candidate_ids=candidate_ids)
                self.apply_gradients(step_ids=ids)
                return loss
        st = self._cstep
        for t in ids:
            ops._chk(t, torch.int64, "ids", 1)
        for t, name in ((sample_weight, "sample_weight"), (candidate_sampling_probability, "candidate_sampling_probability")):
            if t is not None:
                ops._chk(t, torch.float32, name, 1)
        if candidate_ids is not None:
            ops._chk(candidate_ids, torch.int64, "candidate_ids", 1)
        p = ops._p
        for i in range(2):
            for lk in (st.fwd[0][i].lookup, st.bwd[0][i].lookup):
                lk.ids = p(ids[i])
                if i == 1:
                    lk.ids2 = p(category_ids)
            st.tables[i].ids = p(ids[i])
        if st.n_tables == 3:
            st.tables[2].ids = p(ids[2])
        st.dropout_seed = self.dropout_seed
        st.dropout_row0 = self.step_index * self.cfg.batch_size
        st.scorer_precision = ops.SCORER_PRECISIONS.index(self.cfg.scorer_precision)   # (may be switched between steps: bench's second line)
        st.lr, st.eps = self.cfg.learning_rate, self.cfg.adagrad_epsilon                 # (re-read every step, like the Python sequence)
        st.inv_temperature, st.dropout_rate = 1.0 / self.cfg.temperature, self.cfg.dropout_rate
        st.sample_weight, st.cand_prob, st.cand_ids = p(sample_weight), p(candidate_sampling_probability), p(candidate_ids)
        _lib.check(_lib.load().tt_train_step_f32(st, ops._stream()), "tt_train_step_f32")
        self.step_index += 1
        return self.loss

    def evaluate(self, user_ids: torch.Tensor, item_ids: torch.Tensor, **loss_kw) -> torch.Tensor:
        """Forward only (validation loss, SUM over the batch); device tensor, unsynchronised."""
        cfg, ut, it = self.cfg, self.user_tower, self.item_tower
        self._check_categories(loss_kw.get("category_ids"))
        self._check_batch(loss_kw.get("sample_weight"), loss_kw.get("candidate_sampling_probability"),
                          loss_kw.get("candidate_ids"))
        lks = self._lookups(user_ids, item_ids, loss_kw.get("category_ids"))
        if lks is None:
            self._item_inputs(user_ids, item_ids, loss_kw.get("category_ids"))
            q, c = towers_forward(ut, it) if cfg.symmetric else (ut.forward(), it.forward())
        else:
            q, c = towers_forward(ut, it, lookups=lks) if cfg.symmetric else (ut.forward(lookup=lks[0]), it.forward(lookup=lks[1]))
        kw = dict(sample_weight=loss_kw.get("sample_weight"), cand_prob=loss_kw.get("candidate_sampling_probability"),
                  cand_ids=loss_kw.get("candidate_ids"))
        return ops.retrieval_fwd(q, c, 1.0 / cfg.temperature, self.ws, self.lse, self.per_row, self.loss,
                                 precision=cfg.scorer_precision, **kw)

    # ------------------------------------------------------------------ retrieval metrics
    @torch.no_grad()
    def item_corpus_embeddings(self, item_category_ids: torch.Tensor | None = None) -> torch.Tensor:
        """Item-tower output for EVERY item row ([n_items, scorer_dim]), computed batch by batch on the tower's buffers.
        item_category_ids [n_items]: the category bucket of every item (required iff the model has the feature)."""
        it, b = self.item_tower, self.cfg.batch_size
        n = self.cfg.n_items
        if (item_category_ids is None) != (self.cat_table is None):
            raise ValueError("item_category_ids must be given exactly when cfg.n_category_buckets > 0")
        out = torch.empty(n, it.dims[-1], device=self.dev)
        for s in range(0, n, b):
            e = min(s + b, n)
            it.acts[0][:e - s].copy_(self.item_table[s:e])
            if item_category_ids is not None:
                ops.embedding_gather_add_(it.acts[0][:e - s], self.cat_table, item_category_ids[s:e], self.oob)
            it.forward()
            out[s:e].copy_(it.acts[-1][:e - s])
        return out

    @torch.no_grad()
    def evaluate_topk(self, user_ids: torch.Tensor, item_ids: torch.Tensor, metric, corpus: torch.Tensor | None = None):
        """Updates ``metric`` (metrics.FactorizedTopK) with one batch of (user, true item) pairs scored against the
        whole item corpus; returns the ranks."""
        self._check_batch(user_ids, item_ids)
        if corpus is None:
            corpus = self.item_corpus_embeddings()
        q = self.user_tower.forward(lookup=ops.make_lookup(self.user_table, user_ids, oob_flag=self.oob))
        return metric.update_state(q, corpus, item_ids)

    # ------------------------------------------------------------------ checkpoint
    def state_dict(self) -> dict:
        sd = {"config": dict(self.cfg.__dict__), "user_table": self.user_table, "item_table": self.item_table,
              "dense": self.dense_flat, "step_index": self.step_index, "dropout_seed": self.dropout_seed}
        if self.cat_table is not None:
            sd["cat_table"] = self.cat_table
        if self.cfg.optimizer == "adagrad":
            sd.update(user_accum=self.user_accum, item_accum=self.item_accum, dense_accum=self.dense_accum)
            if self.cat_table is not None:
                sd["cat_accum"] = self.cat_accum
        return sd

    def load_state_dict(self, sd: dict):
        for k in ("n_users", "n_items", "embedding_dim", "tower_dims", "item_tower_dims", "optimizer", "n_category_buckets"):
            if sd["config"].get(k, 0 if k == "n_category_buckets" else None) != getattr(self.cfg, k):
                raise ValueError(f"checkpoint {k}={sd['config'].get(k)!r} does not match the trainer's {getattr(self.cfg, k)!r}")
        self.user_table.copy_(sd["user_table"]); self.item_table.copy_(sd["item_table"]); self.dense_flat.copy_(sd["dense"])
        # the counter-based dropout stream continues where the checkpoint stopped (no replayed masks)
        self.step_index = int(sd.get("step_index", 0))
        self.dropout_seed = int(sd.get("dropout_seed", self.dropout_seed))
        if self.cat_table is not None:
            self.cat_table.copy_(sd["cat_table"])
            if self.cfg.optimizer == "adagrad":
                self.cat_accum.copy_(sd["cat_accum"])
        if self.cfg.optimizer == "adagrad":
            self.user_accum.copy_(sd["user_accum"]); self.item_accum.copy_(sd["item_accum"])
            self.dense_accum.copy_(sd["dense_accum"])

    # ------------------------------------------------------------------ HIP graph replay of the whole step
    def capture_graph(self):
        """Capture one train step (its 8-9 launches, one stream) into a HIP graph.  ``step_graph(u, i)`` then
        copies the ids into the captured buffers and replays: one host call per step, no launch gaps."""
        if self.cfg.dropout_rate > 0.0:
            raise NotImplementedError("graph replay with dropout: the per-step counter is a kernel argument")
        b = self.cfg.batch_size
        self._g_uid = torch.zeros(b, dtype=torch.int64, device=self.dev)
        self._g_iid = torch.zeros(b, dtype=torch.int64, device=self.dev)
        self._g_kw = {}
        if self.cat_table is not None:
            self._g_kw["category_ids"] = torch.zeros(b, dtype=torch.int64, device=self.dev)
        cat_state = None if self.cat_table is None else \
            [self.cat_table[:1].clone(), None if self.cat_accum is None else self.cat_accum[:1].clone()]
        state = [t.clone() for t in (self.user_table[:1], self.item_table[:1], self.dense_flat)]
        acc = None
        if self.cfg.optimizer == "adagrad":
            acc = [t.clone() for t in (self.user_accum[:1], self.item_accum[:1], self.dense_accum)]
        # one graph per optimizer path (r04): the skew probe runs OUTSIDE the graphs (step_graph) and picks which one to replay
        poll, load, state_was = self.flag_poll_every, self.range_load, self._skew_state
        self.flag_poll_every = 0                       # (no probe / flag poll inside the warm-up or the capture)
        self._graphs = {}
        try:
            for skewed in ((False, True) if self.skew_limit else (False,)):
                self.range_load = self.skew_limit + 1 if skewed else 0
                self._skew_state = skewed
                s = torch.cuda.Stream(device=self.dev)
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self.step(self._g_uid, self._g_iid, **self._g_kw)   # warm-up on the capture stream (ids 0: touches row 0 only)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.step(self._g_uid, self._g_iid, **self._g_kw)
                torch.cuda.synchronize()
                self._graphs[skewed] = g
                # undo the warm-up step's update (row 0 of both tables, dense parameters)
                self.user_table[:1].copy_(state[0]); self.item_table[:1].copy_(state[1]); self.dense_flat.copy_(state[2])
                if acc is not None:
                    self.user_accum[:1].copy_(acc[0]); self.item_accum[:1].copy_(acc[1]); self.dense_accum.copy_(acc[2])
                if cat_state is not None:
                    self.cat_table[:1].copy_(cat_state[0])
                    if cat_state[1] is not None:
                        self.cat_accum[:1].copy_(cat_state[1])
        finally:
            self.flag_poll_every, self.range_load, self._skew_state = poll, load, state_was
        self._graph = self._graphs[False]
        return self

    def step_graph(self, user_ids: torch.Tensor, item_ids: torch.Tensor, category_ids: torch.Tensor | None = None) -> torch.Tensor:
        self._check_categories(category_ids)
        self._g_uid.copy_(user_ids, non_blocking=True)
        self._g_iid.copy_(item_ids, non_blocking=True)
        ids, rows = [self._g_uid, self._g_iid], [self.cfg.n_users, self.cfg.n_items]
        if category_ids is not None:
            self._g_kw["category_ids"].copy_(category_ids, non_blocking=True)
            ids.append(self._g_kw["category_ids"]); rows.append(self.cfg.n_category_buckets)
        self._replays = getattr(self, "_replays", 0)
        if len(self._graphs) > 1:                      # the skew probe, outside the graphs: which optimizer path this batch replays
            step_was, self.step_index = self.step_index, self._replays
            self._poll_skew(ids, rows)
            self.step_index = step_was
        self._graphs[self._skewed() if len(self._graphs) > 1 else False].replay()
        self._replays += 1
        if self.flag_poll_every and self._replays % self.flag_poll_every == 0:     # host side, outside the graph
            self.poll_ids()
        return self.loss

    def one_launch_optimizer(self, n_ids: int) -> bool:
        """Whether a step of ``n_ids`` pairs would take tt_optimizer_step_ids_f32 (sort + duplicate sums + update in the optimizer
        launch) right now: the shape allows it and the last skew probe saw no overloaded row range."""
        return bool(self.fuse_sort and self.fuse_optimizer and n_ids <= ops.optimizer_ids_max_ids() and not self._skewed())

    def _skewed(self) -> bool:
        if not self.skew_limit:
            return False
        if self.range_load > self.skew_limit:
            self._skew_state = True
        elif 4 * self.range_load < 3 * self.skew_limit:
            self._skew_state = False
        return self._skew_state

    def _poll_skew(self, ids, rows):
        """The skew probe, never waiting for the GPU: takes the result of the probe in flight if it has landed (checked every
        step while one is in flight - a query of an event), and every ``flag_poll_every`` steps starts a new one on this batch's
        ids: one small launch + a 12-byte copy to pinned memory."""
        if not self.skew_limit or not self.fuse_sort or not self.fuse_optimizer or torch.cuda.is_current_stream_capturing():
            return
        if self._skew_event is not None and self._skew_event.query():
            self.range_load, self._skew_event = int(self._skew_host.max().item()), None
        if self._skew_event is None and self.flag_poll_every and self.step_index % self.flag_poll_every == 0:
            if self._skew_dev is None:
                self._skew_dev = torch.zeros(4, dtype=torch.int32, device=self.dev)
                self._skew_host = torch.zeros(4, dtype=torch.int32).pin_memory()
            ops.id_range_load_(self._skew_dev, ids, rows, self.cfg.embedding_dim, self._segs)
            self._skew_host.copy_(self._skew_dev, non_blocking=True)
            self._skew_event = torch.cuda.Event()
            self._skew_event.record()

    def poll_ids(self):
        """Asynchronous check of the out-of-range flag: looks at the copy the PREVIOUS poll started (if it has landed —
        never waits for the GPU) and starts a new 4-byte device-to-pinned-host copy.  step() calls it every
        ``flag_poll_every`` steps; the copy a poll starts is looked at by the NEXT poll, so a bad id is reported within two
        intervals, with the step range it was seen in.  A bad id in the last interval of a run is only seen by check_ids()
        (train.py calls it at the end of every epoch)."""
        if self._oob_event is not None and self._oob_event.query():
            bad, self._oob_event = int(self._oob_host.item()), None
            if bad:
                self.oob.zero_()
                raise IndexError(f"embedding id out of range at or before step {self._oob_step}")
        if self._oob_event is None:
            if self._oob_host is None:
                self._oob_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._oob_host.copy_(self.oob, non_blocking=True)
            self._oob_event = torch.cuda.Event()
            self._oob_event.record()
            self._oob_step = self.step_index

    def check_ids(self):
        """Host check of the out-of-range flag (TF's CPU gather raises InvalidArgumentError); synchronises."""
        self._oob_event = None
        if int(self.oob.item()) != 0:
            self.oob.zero_()
            raise IndexError("embedding id out of range in a previous step")

    def l2_penalty(self) -> torch.Tensor:
        """l2 * sum(W^2) over the Dense kernels — reporting only (the gradient term is fused in the update)."""
        tot = torch.zeros((), device=self.dev, dtype=torch.float64)
        for tower in (self.user_tower, self.item_tower):
            for w in tower.w:
                tot += (w.double() ** 2).sum()
        return self.cfg.l2_regularization * tot
