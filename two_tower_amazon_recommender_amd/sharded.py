"""Row-sharded embedding tables over the GPUs of one node (BASELINE.json configs 4-5).

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI).  Row ``id`` of table t lives on
rank ``id % world`` at local row ``offset_t + id // world`` of that rank's COMBINED shard (mod placement
balances skewed ids; all tables of a rank share one allocation so the owner side is one gather, one sort and
one sparse update per step, and every exchange is ONE collective for all tables).  Per step:

  route   HIP kernel, one workgroup per table: stable partition of the batch by owner into fixed-capacity
          send buffers [world, n_tables, cap] (padding id -1) + the flat slot of every position
                                                                            (tt_route_tables_by_owner_i64)
  C1      all-to-all of the id buffers                      (world*n_tables*cap*8 B per rank)
  K1      owner gathers its rows for the received ids       (HIP gather; -1 -> zero row)
  C2      all-to-all of the rows back                       (world*n_tables*cap*4*dim B per rank)
  ... towers (their first layer reads its input rows straight out of the receive buffer through the routing's flat
      positions - tt_dense_lookup with table = rows_in, ids = pos_flat: no expand gather, no [2B, D] input buffer, r03),
      scorer, loss, every dx of the backward pass ...
  K2'     per-position gradient rows into the send buffer   (tt_scatter_rows_f32, one launch for both tables)
  C3      all-to-all of the gradient rows to the owners     (travels beside the dw GEMMs and the dense reduce)
  K2      owner sorts the received ids, sums the duplicates - inside one rank's batch and across ranks, in (source rank,
          position) order: bitwise reproducible - and applies the fused sparse SGD/Adagrad in ONE launch
          (tt_optimizer_step_ids_f32 on the receive buffer, the dense tower update in the same launch; lists of 16,385 to
          65,536 slots: the long-list kernel of the same entry, r04; longer: plan launch + tt_optimizer_step_f32).  No sort plan on a side stream since r03.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): an all-to-all uses every link at once, one peer per
link; what costs at these sizes (a few MB) is the per-collective latency, hence three all-to-alls per step
for all tables together instead of three per table.  The buffers are fixed-size so no step waits on the host;
``capacity_factor`` x the mean positions per peer is reserved and an overflow raises at the next ``check()``.

The exchange code is device-agnostic torch.distributed; the row kernels come from a ``backend`` (default:
the HIP ops; the CPU/gloo tests pass a NumPy-oracle backend defined in the test).
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.distributed as dist


class HipRowBackend:
    """The product's row kernels (C ABI of include/twotower_hip.h).  No fallback."""

    def __init__(self, device):
        from . import ops
        self.ops = ops
        self.device = device
        self._plans = {}          # (owner key, n ids) -> SparsePlan; the key is the ShardedTables instance using the backend
        # High-priority streams: ROCm keeps a separate hardware-queue pool per priority, so these never land on the
        # main stream's queue (with the default priority the sort plan ended up serialised behind the compute kernels
        # whenever the stream -> queue round-robin happened to collide, e.g. after RCCL had created its streams)
        self._side = torch.cuda.Stream(device=device, priority=-1)
        self._pre = torch.cuda.Stream(device=device, priority=-1)
        self._pending = {}        # owner key -> the plan its next apply() consumes

    @contextlib.contextmanager
    def prefetch_stream(self):
        """Work issued inside runs on a side stream, after everything queued so far on the current stream."""
        self._pre.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._pre):
            yield

    def join_prefetch(self):
        torch.cuda.current_stream().wait_stream(self._pre)

    def route(self, ids_list, world, num_rows_list, offsets, cap, send_ids, pos_flats, flags):
        self.ops.route_tables_by_owner(ids_list, world, num_rows_list, offsets, cap, send_ids, pos_flats, flags)

    def gather(self, table, ids, out, oob_flag):
        self.ops.embedding_gather(table, ids, out=out, oob_flag=oob_flag)

    def scatter_rows(self, src, idx, dst):
        self.ops.scatter_rows(src, idx, dst)

    def plan(self, ids, num_rows, key=None):
        """Sort the owner-side ids on a side stream (they are known right after C1).  ``key`` names the caller
        (one ShardedTables): several sharded tables may share a backend without sharing plans."""
        n = ids.numel()
        plan = self._plans.get((key, n, ids.data_ptr()))
        if plan is None:
            plan = self._plans[(key, n, ids.data_ptr())] = self.ops.SparsePlan(n, ids.device)
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            plan.run(ids, num_rows)
        self._pending[key] = plan

    @property
    def max_fused_ids(self) -> int:
        """Longest owner-side id list the one-launch optimizer (sort + duplicate sums + update) takes."""
        return self.ops.optimizer_ids_max_ids()

    def range_load(self, out, ids, num_rows, dim, dense_segs):
        """Skew probe (tt_id_range_load): out[0] = the most owner-side ids in one row range of apply_ids' launch."""
        self.ops.id_range_load_(out, [ids], [num_rows], dim, dense_segs)

    def apply_ids(self, opt, table, accum, ids, grads, dense_segs, lr, eps, key=None):
        """Owner side in ONE launch, straight from the received ids (tt_optimizer_step_ids_f32): the workgroups sort the
        ids of their row range in LDS, sum the duplicate gradient rows - inside one rank's batch and across ranks, in
        (source rank, position) order - and apply the update; the dense segments are updated beside them.  No plan launch,
        no side stream.  Bit-identical to plan() + apply() + dense_update_."""
        n = ids.numel()
        ws = self._plans.get((key, n, "ws"))
        if ws is None:
            ws = self._plans[(key, n, "ws")] = self.ops.SparsePlan(n, ids.device)      # (lends its apply workspace)
        self.ops.optimizer_step_ids_(opt, [(table, accum, grads, ids, ws)], dense_segs, lr, eps)

    def apply(self, opt, table, accum, ids, grads, lr, eps, key=None):
        plan = self._pending.pop(key)
        torch.cuda.current_stream().wait_stream(self._side)
        if opt == "sgd":
            self.ops.sparse_sgd_(table, grads, plan, lr)
        else:
            self.ops.sparse_adagrad_(table, accum, grads, plan, lr, eps)


class _Ready:
    """Stand-in for a CUDA event on the CPU/gloo test path: the copy is synchronous, so it is always complete."""

    @staticmethod
    def query():
        return True


def shard_rows(num_rows: int, world: int, rank: int) -> int:
    """Rows of a table that live on ``rank`` (ids rank, rank+world, ...)."""
    return (num_rows - rank + world - 1) // world if num_rows > rank else 0


class ShardedTables:
    """Several row-sharded tables of one width behind one set of exchange buffers (see the module docstring)."""

    def __init__(self, num_rows, dim: int, batch: int, device, group=None, capacity_factor: float = 2.0,
                 backend=None, table: torch.Tensor | None = None, accum: torch.Tensor | None = None,
                 force_collectives: bool = False, sync_ops_inline: bool = True):
        self.group = group
        # force_collectives: issue the collectives even in a one-rank group (exercises the RCCL calls on one GPU)
        self.collectives = force_collectives and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.num_rows = [int(n) for n in num_rows]
        self.n_tables = nt = len(self.num_rows)
        self.dim, self.batch, self.device = dim, batch, device
        w = self.world
        mean = (batch + w - 1) // w
        cap = batch if w == 1 else min(batch, int(mean * capacity_factor + 63) // 64 * 64)
        self.cap = max(cap, 1)
        # every rank reserves ceil(rows/world) rows per table, so local offsets are the same on all ranks
        self.rows_cap = [max((n + w - 1) // w, 1) for n in self.num_rows]
        self.offsets = [sum(self.rows_cap[:t]) for t in range(nt)]
        total = sum(self.rows_cap)
        if table is not None and table.shape[0] < self.offsets[-1] + max(shard_rows(self.num_rows[-1], w, self.rank), 1):
            raise ValueError("ShardedTables: the given shard has too few rows")
        self.table = table if table is not None else torch.zeros(total, dim, device=device)
        self.accum = accum
        self.backend = backend if backend is not None else HipRowBackend(device)
        n = w * nt * self.cap
        self.collectives = self.collectives or w > 1
        # sync_ops_inline=False: issue every exchange as an asynchronous op (the behaviour before r01h)
        self.sync_ops_inline = sync_ops_inline
        self.rows_out = torch.empty(n, dim, device=device)        # owner side: gathered rows / received grads
        # requester side: received rows / grads to send (one rank without collectives: the exchange is the identity)
        self.rows_in = torch.empty(n, dim, device=device) if self.collectives else self.rows_out
        # gradient rows travel in buffers of their own: the received embedding rows stay valid through the backward pass (the
        # towers' first layer reads them in place - forward GEMM and dW GEMM - through pos_flat: no expanded [batch, dim] copy)
        self.grad_send = torch.empty(n, dim, device=device)
        self.grad_recv = torch.empty(n, dim, device=device) if self.collectives else self.grad_send
        # owner side in one launch from the received ids (no sort plan) when the backend offers it and the list fits
        self.fused_apply = hasattr(self.backend, "apply_ids") and n <= getattr(self.backend, "max_fused_ids", 0)
        # r04, as in TwoTowerTrainer: apply_ids gives every row range of the combined shard to one workgroup, and a vocabulary in
        # order of frequency crowds the first ranges of every owner (local row = id // world keeps the order).  While the lagged
        # probe (backend.range_load every probe_every steps on the received ids, read from pinned memory when it has landed)
        # sees a range of more than skew_limit ids, the owner side takes plan() + apply().  Each rank decides for itself: the
        # collectives of the step are the same on both paths.  probe_segs: the dense segments that ride in apply_ids' launch
        # (they set its row ranges); None = no probing.
        self.skew_limit, self.probe_every, self.probe_segs = int(os.environ.get("TT_SKEW_LIMIT", "384")), 50, None
        self.range_load = 0
        self._skew_state = False
        self._fused_now = self.fused_apply
        self._skew_dev = self._skew_host = self._skew_event = None
        self._lookups = 0
        # id buffers exist twice: lookup_prefetch() routes and exchanges the NEXT step's ids while this step computes
        self._idbufs = [self._make_idbufs(), None]
        self._cur = 0
        self._prefetched = None
        self.flags = torch.zeros(2, dtype=torch.int32, device=device)   # [oob, overflow]
        self._flag_host = None          # pinned copy of the flags taken by poll(); checked one poll later, without a sync
        self._flag_event = None
        self._flag_step = -1

    def _make_idbufs(self):
        n = self.world * self.n_tables * self.cap
        i64 = dict(dtype=torch.int64, device=self.device)
        send = torch.empty(n, **i64)
        recv = torch.empty(n, **i64) if self.collectives else send
        pos = torch.empty(self.n_tables * self.batch, **i64)      # table t's positions at [t*batch, (t+1)*batch)
        return send, recv, pos, [pos[t * self.batch:(t + 1) * self.batch] for t in range(self.n_tables)]

    send_ids = property(lambda self: self._idbufs[self._cur][0])
    recv_ids = property(lambda self: self._idbufs[self._cur][1])
    pos_flat = property(lambda self: self._idbufs[self._cur][2])
    pos_flats = property(lambda self: self._idbufs[self._cur][3])

    def shard(self, t: int) -> torch.Tensor:
        """This rank's rows of table t (a view of the combined shard)."""
        return self.table[self.offsets[t]:self.offsets[t] + shard_rows(self.num_rows[t], self.world, self.rank)]

    def accum_shard(self, t: int) -> torch.Tensor:
        return self.accum[self.offsets[t]:self.offsets[t] + shard_rows(self.num_rows[t], self.world, self.rank)]

    def _a2a(self, out, inp, overlap: bool = True):
        """All-to-all; returns a work handle to ``wait()`` on, or None.
        overlap=True: asynchronous (the collective runs on the backend's own stream; ``wait()`` makes the current
        stream wait for it without blocking the host, so kernels issued before the wait overlap the transfer).
        overlap=False: for an exchange whose result is needed at once — issued as a synchronous op, which
        ProcessGroupNCCL (torch >= 2.7) enqueues on the CURRENT stream: no event hand-offs to and from a second
        stream (each costs the GPU queue 4-7 us); the host is not blocked either way."""
        if not self.collectives:
            return None
        if overlap or not self.sync_ops_inline:
            return dist.all_to_all_single(out, inp, group=self.group, async_op=True)
        dist.all_to_all_single(out, inp, group=self.group, async_op=False)
        return None

    @staticmethod
    def _wait(work):
        if work is not None:
            work.wait()

    # ---------------------------------------------------------------- forward, in three phases
    def lookup_start(self, ids_list):
        """route + C1 (ids to their owners); ids_list[t] = this rank's ids into table t (all of length batch).
        If exactly these tensors were handed to lookup_prefetch() the exchange is already on its way."""
        ids_list = list(ids_list)
        pre, self._prefetched = self._prefetched, None
        if pre is not None and len(pre[0]) == len(ids_list) and all(a is b for a, b in zip(pre[0], ids_list)):
            self._cur = 1 - self._cur
            self.backend.join_prefetch()
            # routed ahead of time; the id exchange is either already on its way or starts now
            self._w = pre[1] if pre[2] else self._a2a(self.recv_ids, self.send_ids, overlap=False)
            return
        if pre is not None:                       # prefetched for other ids: let that exchange finish, then ignore it
            self._wait(pre[1])
            self.backend.join_prefetch()
        self.backend.route(ids_list, self.world, self.num_rows, self.offsets, self.cap, self.send_ids, self.pos_flats,
                           self.flags)
        self._w = self._a2a(self.recv_ids, self.send_ids, overlap=False)           # C1

    def lookup_prefetch(self, ids_list, exchange: bool = False):
        """Route the NEXT step's ids (they do not depend on the tables) into the other id-buffer set on a side stream,
        so the single-workgroup routing kernel (~12 us) leaves the next step's critical path.  exchange=True also
        issues their id all-to-all (C1) there; on one rank that costs more than it hides (the collective's kernel
        disturbs the scorer), so it is off by default.  Call it after lookup_finish()."""
        ids_list = list(ids_list)
        nxt = 1 - self._cur
        if self._idbufs[nxt] is None:
            self._idbufs[nxt] = self._make_idbufs()
        send, recv, _, pos_flats = self._idbufs[nxt]
        work = None
        with self.backend.prefetch_stream():
            self.backend.route(ids_list, self.world, self.num_rows, self.offsets, self.cap, send, pos_flats, self.flags)
            if exchange:
                work = self._a2a(recv, send)                                       # C1 of the next step
        self._prefetched = (ids_list, work, exchange)

    def lookup_rows(self):
        """owner side: sort plan (side stream), K1 gather, C2 (rows back to the requesters)."""
        self._wait(self._w)
        be = self.backend
        self._poll_skew()
        if self.skew_limit and self.range_load > self.skew_limit:
            self._skew_state = True
        elif 4 * self.range_load < 3 * self.skew_limit:
            self._skew_state = False
        self._fused_now = self.fused_apply and not self._skew_state
        if not self._fused_now:
            be.plan(self.recv_ids, self.table.shape[0], key=id(self))
        be.gather(self.table, self.recv_ids, self.rows_out, self.flags[0:1])       # K1
        self._w = self._a2a(self.rows_in, self.rows_out, overlap=False)            # C2

    def _poll_skew(self):
        """Never waits for the GPU: takes the probe in flight if it has landed, starts a new one every ``probe_every`` lookups."""
        be = self.backend
        if not self.fused_apply or self.probe_segs is None or not self.skew_limit or not hasattr(be, "range_load"):
            return
        if self._skew_event is not None and self._skew_event.query():
            self.range_load, self._skew_event = int(self._skew_host[0]), None
        if self._skew_event is None and self.probe_every and self._lookups % self.probe_every == 0:
            if self._skew_dev is None:
                self._skew_dev = torch.zeros(4, dtype=torch.int32, device=self.device)
                self._skew_host = torch.zeros(4, dtype=torch.int32).pin_memory()
            be.range_load(self._skew_dev, self.recv_ids, self.table.shape[0], self.table.shape[1], self.probe_segs)
            self._skew_host.copy_(self._skew_dev, non_blocking=True)
            self._skew_event = torch.cuda.Event()
            self._skew_event.record()
        self._lookups += 1

    def lookup_wait(self):
        """The received rows are in ``rows_in`` (row of position p of table t at slot pos_flats[t][p], -1 = none): a consumer
        that indexes them itself - the towers' first layer, tt_dense_lookup - needs no expanded copy."""
        self._wait(self._w)
        self._w = None

    def lookup_finish(self, out: torch.Tensor):
        """K1': out[t*batch + p, :] = row of position p of table t."""
        self._wait(self._w)
        self._w = None
        self.backend.gather(self.rows_in, self.pos_flat, out, None)
        return out

    def lookup(self, ids_list, out: torch.Tensor):
        self.lookup_start(ids_list)
        self.lookup_rows()
        return self.lookup_finish(out)

    # ---------------------------------------------------------------- backward, in two phases
    def grads_start(self, grads: torch.Tensor):
        """K2' + C3: per-position gradient rows [n_tables*batch, dim] to the owners (padding slots are never read)."""
        self.backend.scatter_rows(grads, self.pos_flat, self.grad_send)
        self._w = self._a2a(self.grad_recv, self.grad_send)                        # C3

    def grads_finish(self, opt: str, lr: float, eps: float = 1e-7, dense_segs=None):
        """K2: fused sparse update on the owner (duplicates summed first, in (source rank, position) order).
        dense_segs (only with ``fused_apply``): dense segments updated in the same launch; returns True if they were."""
        self._wait(self._w)
        if self._fused_now and dense_segs:
            self.backend.apply_ids(opt, self.table, self.accum, self.recv_ids, self.grad_recv, dense_segs, lr, eps, key=id(self))
            return True
        if self._fused_now:        # (no dense work to ride along: the sort plan was skipped, so run it now)
            self.backend.plan(self.recv_ids, self.table.shape[0], key=id(self))
        self.backend.apply(opt, self.table, self.accum, self.recv_ids, self.grad_recv, lr, eps, key=id(self))
        return False

    def apply_gradients(self, grads: torch.Tensor, opt: str, lr: float, eps: float = 1e-7):
        """grads[t*batch + p, :] = dLoss/d(out[t*batch + p, :]) of the last lookup (K2', C3, K2)."""
        self.grads_start(grads)
        self.grads_finish(opt, lr, eps)

    def _raise_flags(self, f, when: str):
        if int(f[0]):
            raise IndexError(f"embedding id out of range {when}")
        if int(f[1]):
            raise RuntimeError(f"sharded exchange overflow {when}: more than {self.cap} positions for one owner; "
                               "raise capacity_factor")

    def poll(self, step_index: int):
        """Asynchronous check of [out-of-range, overflow]: looks at the flag copy taken by the PREVIOUS poll (if its
        8-byte device-to-pinned-host copy has landed — never waits for the GPU) and starts a new one.  A bad id or an
        overflowed bucket is therefore reported one polling interval late at most, with the step range it happened in,
        instead of at the end of the epoch."""
        if self._flag_event is not None and self._flag_event.query():
            f = self._flag_host.tolist()
            self._flag_event = None
            if int(f[0]) or int(f[1]):
                self.flags.zero_()
                self._raise_flags(f, f"at or before step {self._flag_step}")
        if self._flag_event is None:
            if self._flag_host is None:
                self._flag_host = torch.zeros(2, dtype=torch.int32).pin_memory() if self.flags.is_cuda \
                    else torch.zeros(2, dtype=torch.int32)
            self._flag_host.copy_(self.flags, non_blocking=True)
            if self.flags.is_cuda:
                self._flag_event = torch.cuda.Event()
                self._flag_event.record()
            else:
                self._flag_event = _Ready()
            self._flag_step = step_index

    def check(self):
        """Host check (synchronises): out-of-range ids (TF's gather raises) and exchange-buffer overflow."""
        f = self.flags.tolist()          # (a copy: .cpu() would alias a CPU tensor)
        self.flags.zero_()
        self._flag_event = None
        if int(f[0]):
            raise IndexError("embedding id out of range in a previous step")
        if int(f[1]):
            raise RuntimeError(f"sharded exchange overflow: more than {self.cap} positions for one owner; "
                               "raise capacity_factor")


class ShardedEmbedding(ShardedTables):
    """One row-sharded table: ``lookup(ids, out)`` / ``apply_gradients(grads, ...)`` on this rank's batch."""

    def __init__(self, num_rows: int, dim: int, batch: int, device, group=None, capacity_factor: float = 2.0,
                 backend=None, table: torch.Tensor | None = None, accum: torch.Tensor | None = None):
        super().__init__([num_rows], dim, batch, device, group, capacity_factor, backend, table, accum)

    def lookup_start(self, ids):
        super().lookup_start([ids])

    def lookup_prefetch(self, ids, exchange: bool = False):
        super().lookup_prefetch([ids], exchange)

    def lookup(self, ids, out):
        self.lookup_start(ids)
        self.lookup_rows()
        return self.lookup_finish(out)


class ShardedTwoTowerTrainer:
    """Data-parallel towers + row-sharded tables: one instance per rank / GPU.

    ``cfg.batch_size`` is the PER-RANK batch; ``cfg.n_users`` / ``cfg.n_items`` are the GLOBAL row counts.
    negatives="local"  — each rank's queries score against its own candidates (what tfrs.tasks.Retrieval
                         does under a data-parallel tf.distribute strategy: the task sees the per-replica batch);
                         total loss = sum over ranks.
    negatives="global" — candidates (and their ids) are all-gathered, every query sees world*batch candidates
                         (identical to the single-device loss on the global batch), dC is reduce-scattered.
    Dense tower gradients are summed with one all-reduce of a flat ~0.5 MB bucket.  The hashed category table
    (cfg.n_category_buckets rows; BASELINE configs[4]) is tiny, so it is REPLICATED like the dense parameters: every
    rank de-duplicates its own gradient rows into a [buckets, dim] matrix at the tail of that bucket, the all-reduce
    sums it, and the dense update applies it (rows no rank touched get a zero gradient: unchanged).
    """

    def __init__(self, cfg, device, group=None, seed: int | None = None, negatives: str = "local",
                 capacity_factor: float = 2.0, force_collectives: bool = False, sync_ops_inline: bool = True):
        from . import ops
        from .trainer import Tower, TID_USER_TABLE, TID_ITEM_TABLE
        cfg.validate()
        if negatives not in ("local", "global"):
            raise ValueError("negatives must be 'local' or 'global'")
        self.ops, self.cfg, self.group, self.negatives = ops, cfg, group, negatives
        self.dev = dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("ShardedTwoTowerTrainer needs a CUDA/HIP device: there is no CPU fallback")
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        b, d, w = cfg.batch_size, cfg.embedding_dim, self.world
        adagrad = cfg.optimizer == "adagrad"
        # both tables behind one set of exchange buffers: 3 all-to-alls per step, one owner-side gather/sort/update
        self.emb = ShardedTables([cfg.n_users, cfg.n_items], d, b, dev, group, capacity_factor,
                                 force_collectives=force_collectives, sync_ops_inline=sync_ops_inline)
        self.collectives = self.emb.collectives
        if adagrad:
            self.emb.accum = torch.full_like(self.emb.table, cfg.adagrad_initial_accumulator)
        n_tower = Tower.param_count(cfg, cfg.user_dims)
        n_item = Tower.param_count(cfg, cfg.item_dims)
        n_cat = cfg.n_category_buckets * d
        self.dense_flat = torch.zeros(n_tower + n_item + n_cat, device=dev)
        self.dense_accum = torch.full_like(self.dense_flat, cfg.adagrad_initial_accumulator) if adagrad else None
        self.dense_grad = torch.empty_like(self.dense_flat)
        self.user_tower = Tower(cfg, cfg.user_dims, self.dense_flat, self.dense_accum, 0, dev)
        self.item_tower = Tower(cfg, cfg.item_dims, self.dense_flat, self.dense_accum, n_tower, dev)
        self.cat_table = self.cat_accum = self.cat_grad = self.cat_plan = None
        if n_cat:
            nb = cfg.n_category_buckets
            self.cat_table = self.dense_flat[n_tower + n_item:].view(nb, d)
            self.cat_grad = self.dense_grad[n_tower + n_item:].view(nb, d)
            self.cat_accum = self.dense_accum[n_tower + n_item:].view(nb, d) if adagrad else None
            self.cat_plan = ops.SparsePlan(b, dev)
        # the towers' inputs / input gradients are the two halves of one buffer: one expand gather, one scatter
        self.emb_in = torch.empty(2 * b, d, device=dev)
        self.emb_grad = torch.empty(2 * b, d, device=dev)
        self.user_tower.acts[0], self.item_tower.acts[0] = self.emb_in[:b], self.emb_in[b:]
        self.user_tower.demb, self.item_tower.demb = self.emb_grad[:b], self.emb_grad[b:]
        l2 = cfg.l2_regularization
        # pass 1: slabs -> flat gradient (no update); pass 2 (after the all-reduce): update from the flat gradient
        self._segs_reduce = self.user_tower.segments(l2, self.dense_grad, 0) + \
            self.item_tower.segments(l2, self.dense_grad, n_tower)
        self._segs_apply = []
        off = 0
        for tower in (self.user_tower, self.item_tower):
            for l in range(tower.n_layers):
                for p, acc, reg in ((tower.w[l], tower.w_acc[l], l2), (tower.b[l], tower.b_acc[l], 0.0)):
                    self._segs_apply.append(ops.make_dense_seg(p, acc, self.dense_grad[off:off + p.numel()], 1, reg))
                    off += p.numel()
        if n_cat:
            self._segs_apply.append(ops.make_dense_seg(self.cat_table, self.cat_accum, self.cat_grad, 1, 0.0))
        # the owner-side skew probe cuts the combined shard like the launch it stands in for: by the dense segments riding in it
        self.emb.probe_segs = self._segs_apply if self.collectives else self._segs_reduce
        sd = cfg.tower_dims[-1]
        nc = b * w if negatives == "global" else b
        self.ws = torch.empty(ops.retrieval_workspace_bytes(b, nc, sd), dtype=torch.uint8, device=dev)
        self.lse = torch.empty(b, device=dev)
        self.per_row = torch.empty(b, device=dev)
        self.loss = torch.empty(1, device=dev)
        if negatives == "global":
            self.c_all = torch.empty(nc, sd, device=dev)
            self.dc_all = torch.empty(nc, sd, device=dev)
        self.step_index = 0
        self.fuse_lookup = True       # the towers' first layer reads the received rows in place (False: expand gather + emb_in)
        self.flag_poll_every = 50     # steps between asynchronous [out-of-range, overflow] flag polls (0 = never)
        self.dropout_seed = 0 if seed is None else seed
        if seed is not None:
            self.init_synthetic(seed)

    def set_negatives(self, negatives: str):
        """Switch between global and local in-batch negatives (the scorer workspace of the other slab shape is allocated
        on first use; the exchange, the towers and the optimizers are the same)."""
        if negatives not in ("local", "global"):
            raise ValueError("negatives must be 'local' or 'global'")
        if negatives == self.negatives:
            return
        b, w, sd = self.cfg.batch_size, self.world, self.cfg.tower_dims[-1]
        nc = b * w if negatives == "global" else b
        self._ws_by_mode = getattr(self, "_ws_by_mode", {self.negatives: self.ws})
        if negatives not in self._ws_by_mode:
            self._ws_by_mode[negatives] = torch.empty(self.ops.retrieval_workspace_bytes(b, nc, sd), dtype=torch.uint8, device=self.dev)
        self.ws = self._ws_by_mode[negatives]
        if negatives == "global" and not hasattr(self, "c_all"):
            self.c_all = torch.empty(nc, sd, device=self.dev)
            self.dc_all = torch.empty(nc, sd, device=self.dev)
        self.negatives = negatives

    def init_synthetic(self, seed: int):
        """Same values as the single-GPU trainer / oracle.synthetic_state: each rank fills only its rows."""
        import math
        from .trainer import TID_USER_TABLE, TID_ITEM_TABLE, TID_DENSE_BASE, TID_CATEGORY_TABLE
        ops, w, r = self.ops, self.world, self.rank
        for t, tid in enumerate((TID_USER_TABLE, TID_ITEM_TABLE)):
            shard = self.emb.shard(t)
            if shard.shape[0]:
                ops.fill_uniform_rows_(shard, seed, tid, -0.05, 0.1, row_start=r, row_stride=w)
        self.dense_flat.zero_()
        for t, tower in enumerate((self.user_tower, self.item_tower)):
            for l, wt in enumerate(tower.w):
                lim = torch.tensor(math.sqrt(6.0 / (wt.shape[0] + wt.shape[1])), dtype=torch.float64).to(torch.float32)
                ops.fill_uniform_(wt, seed, TID_DENSE_BASE + 2 * l + t, -lim.item(), (lim + lim).item())
        if self.cat_table is not None:
            ops.fill_uniform_(self.cat_table, seed, TID_CATEGORY_TABLE, -0.05, 0.1)

    def synthetic_batch(self, seed: int, step: int, variant: str = "U", out=None):
        """This rank's slice of the global synthetic batch of step ``step`` (global batch = world * batch)."""
        from .trainer import TID_USER_IDS, TID_ITEM_IDS
        b = self.cfg.batch_size
        if out is None:
            out = (torch.empty(b, dtype=torch.int64, device=self.dev), torch.empty(b, dtype=torch.int64, device=self.dev))
        start = (step * self.world + self.rank) * b
        self.ops.fill_ids_(out[0], seed, TID_USER_IDS, self.cfg.n_users, variant, start=start)
        self.ops.fill_ids_(out[1], seed, TID_ITEM_IDS, self.cfg.n_items, variant, start=start)
        return out

    def synthetic_categories(self, seed: int, step: int, variant: str = "Z", out=None):
        """This rank's slice of the category buckets of global synthetic step ``step``."""
        from .trainer import TID_CATEGORY_IDS
        b = self.cfg.batch_size
        if out is None:
            out = torch.empty(b, dtype=torch.int64, device=self.dev)
        self.ops.fill_ids_(out, seed, TID_CATEGORY_IDS, self.cfg.n_category_buckets, variant,
                           start=(step * self.world + self.rank) * b)
        return out

    def _cand_prob(self, p):
        """candidate_sampling_probability of the candidates this rank scores against (all-gathered for global negatives)."""
        if p is None or self.negatives == "local" or not self.collectives:
            return p
        if not hasattr(self, "p_all"):
            self.p_all = torch.empty(self.cfg.batch_size * self.world, device=self.dev)
        dist.all_gather_into_tensor(self.p_all, p.contiguous(), group=self.group)
        return self.p_all

    def _cand_ids(self, ids):
        """candidate_ids (accidental-hit removal) of the candidates this rank scores against: its own batch's for local
        negatives, every rank's (all-gathered, rank-major like the candidate embeddings) for global negatives."""
        if ids is None or self.negatives == "local" or not self.collectives:
            return ids
        if not hasattr(self, "id_all"):
            self.id_all = torch.empty(self.cfg.batch_size * self.world, dtype=torch.int64, device=self.dev)
        dist.all_gather_into_tensor(self.id_all, ids.contiguous(), group=self.group)
        return self.id_all

    def step(self, user_ids: torch.Tensor, item_ids: torch.Tensor, next_ids=None, category_ids=None,
             candidate_sampling_probability=None, prefetch_exchange: bool = False, sample_weight=None,
             candidate_ids=None) -> torch.Tensor:
        """One train step on this rank's batch; returns this rank's (device, unsynchronised) loss.
        sample_weight [batch] weighs this rank's pairs; candidate_sampling_probability / candidate_ids [batch] describe
        this rank's candidates (logQ correction / accidental-hit removal, tfrs.tasks.Retrieval) and are all-gathered
        with the candidates when negatives are global.
        next_ids = (user_ids, item_ids) of the following step, if the input pipeline already has them: they are
        routed beside this step's scorer (and, with prefetch_exchange, their id all-to-all is issued there too); pass
        the same tensors to the next call."""
        from .trainer import towers_forward, towers_backward
        cfg, ops, ut, it, em = self.cfg, self.ops, self.user_tower, self.item_tower, self.emb
        b, w = cfg.batch_size, self.world
        if (category_ids is None) != (self.cat_table is None):
            raise ValueError("category_ids must be given exactly when cfg.n_category_buckets > 0")
        if user_ids.numel() != b or item_ids.numel() != b:
            raise ValueError(f"batch must have {b} pairs per rank")
        for name, v in (("sample_weight", sample_weight), ("candidate_sampling_probability", candidate_sampling_probability),
                        ("candidate_ids", candidate_ids)):
            if v is not None and v.numel() != b:
                raise ValueError(f"{name} must have {b} entries (this rank's batch)")
        if self.flag_poll_every and self.step_index % self.flag_poll_every == 0:
            em.poll(self.step_index)              # bad ids / overflowed buckets surface within one interval, no sync
        em.lookup_start((user_ids, item_ids))
        em.lookup_rows()
        if category_ids is not None:          # the replicated table's sort plan joins the owner plan on the side stream
            side = em.backend._side
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.cat_plan.run(category_ids, cfg.n_category_buckets)
        # K1' fused into the towers' first layer (r03): the forward GEMM and the dW GEMM read the received rows in place
        # through pos_flat (tt_dense_lookup with table = rows_in; the category row is its table2) - no expand gather, no
        # [2*batch, dim] emb_in, no gather-add launch.  (Longer batches than the fused lookup takes: the old three launches.)
        fused_in = self.fuse_lookup and b <= ops.MAX_FUSED_LOOKUP_ROWS
        lks = None
        if fused_in:
            em.lookup_wait()
            lks = (ops.make_lookup(em.rows_in, em.pos_flats[0], oob_flag=em.flags[0:1]),
                   ops.make_lookup(em.rows_in, em.pos_flats[1], self.cat_table, category_ids, em.flags[0:1]))
        else:
            em.lookup_finish(self.emb_in)
            if category_ids is not None:
                ops.embedding_gather_add_(it.acts[0], self.cat_table, category_ids, em.flags[0:1])
        if next_ids is not None:
            em.lookup_prefetch(next_ids, prefetch_exchange)
        row0 = (self.step_index * w + self.rank) * b          # first global batch row of this rank
        if cfg.symmetric:
            q, c = towers_forward(ut, it, (cfg.dropout_rate, self.dropout_seed, row0), lookups=lks)
        else:
            q = ut.forward((cfg.dropout_rate, self.dropout_seed, 0, row0), lookup=lks[0] if lks else None)
            c = it.forward((cfg.dropout_rate, self.dropout_seed, 1, row0), lookup=lks[1] if lks else None)
        inv_t = 1.0 / cfg.temperature
        cp = self._cand_prob(candidate_sampling_probability)
        ci = self._cand_ids(candidate_ids)
        if self.negatives == "local" or not self.collectives:
            ops.retrieval_fwd_bwd(q, c, inv_t, self.ws, self.lse, self.per_row, self.loss, ut.dz[-1], it.dz[-1], cand_prob=cp,
                                  sample_weight=sample_weight, cand_ids=ci, precision=cfg.scorer_precision)
        else:
            dist.all_gather_into_tensor(self.c_all, c, group=self.group)                       # C4
            off = self.rank * b
            ops.retrieval_fwd_bwd(q, self.c_all, inv_t, self.ws, self.lse, self.per_row, self.loss, ut.dz[-1], self.dc_all,
                                  diag_offset=off, cand_prob=cp, sample_weight=sample_weight, cand_ids=ci,
                                  precision=cfg.scorer_precision)
            dist.reduce_scatter_tensor(it.dz[-1], self.dc_all, op=dist.ReduceOp.SUM, group=self.group)   # C5

        # every dx first: the embedding gradient rows travel to their owners beside the dw GEMMs and the dense reduce
        def send():
            em.grads_start(self.emb_grad)
            if category_ids is not None and self.collectives:
                # this rank's de-duplicated category gradient rows: 0 - (-1 * g_sum) = g_sum exactly
                torch.cuda.current_stream().wait_stream(em.backend._side)
                self.cat_grad.zero_()
                ops.sparse_sgd_(self.cat_grad, it.demb, self.cat_plan, -1.0)
        if cfg.symmetric and not self.collectives:
            # nothing to overlap the dw GEMMs with: the fused dx + dw launches of the plain trainer, then the scatter
            towers_backward(ut, it, cfg.dropout_rate, lookups=lks)
            send()
        elif cfg.symmetric:
            towers_backward(ut, it, cfg.dropout_rate, on_embedding_grads=send, lookups=lks)
        else:
            ut.backward(cfg.dropout_rate, dx=True, dw=False)
            it.backward(cfg.dropout_rate, dx=True, dw=False)
            send()
            ut.backward(cfg.dropout_rate, dx=False, dw=True, lookup=lks[0] if lks else None)
            it.backward(cfg.dropout_rate, dx=False, dw=True, lookup=lks[1] if lks else None)
        self.step_index += 1
        if not self.collectives:
            # one rank, no collectives: owner sort + duplicate sums + sparse update + dense update in ONE launch
            if not em.grads_finish(cfg.optimizer, cfg.learning_rate, cfg.adagrad_epsilon, dense_segs=self._segs_reduce):
                ops.dense_update_(self._segs_reduce, cfg.optimizer, cfg.learning_rate, cfg.adagrad_epsilon, apply=True)
            if category_ids is not None:      # one rank, no collectives: the plain trainer's sparse update
                if cfg.optimizer == "sgd":
                    ops.sparse_sgd_(self.cat_table, it.demb, self.cat_plan, cfg.learning_rate)
                else:
                    ops.sparse_adagrad_(self.cat_table, self.cat_accum, it.demb, self.cat_plan, cfg.learning_rate,
                                        cfg.adagrad_epsilon)
        else:
            ops.dense_update_(self._segs_reduce, cfg.optimizer, cfg.learning_rate, cfg.adagrad_epsilon, apply=False)
            if em.sync_ops_inline and em._fused_now:
                # C6 on the current stream, then ONE launch: the owner's sort + duplicate sums + sparse update (it waits
                # for C3) with the dense update from the all-reduced gradient riding along
                dist.all_reduce(self.dense_grad, op=dist.ReduceOp.SUM, group=self.group)                   # C6
                em.grads_finish(cfg.optimizer, cfg.learning_rate, cfg.adagrad_epsilon, dense_segs=self._segs_apply)
                return self.loss
            if em.sync_ops_inline:
                # C6 on the current stream, after the owner update (which waits for C3): the ~12 us of sparse update it
                # could have overlapped are less than the two cross-stream hand-offs of an asynchronous op cost
                em.grads_finish(cfg.optimizer, cfg.learning_rate, cfg.adagrad_epsilon)
                dist.all_reduce(self.dense_grad, op=dist.ReduceOp.SUM, group=self.group)                   # C6
            else:
                ar = dist.all_reduce(self.dense_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                em.grads_finish(cfg.optimizer, cfg.learning_rate, cfg.adagrad_epsilon)
                ar.wait()
            ops.dense_update_(self._segs_apply, cfg.optimizer, cfg.learning_rate, cfg.adagrad_epsilon, apply=True)
        return self.loss

    @property
    def user_table(self) -> torch.Tensor:
        """This rank's rows of the user table (global rows rank, rank+world, ...)."""
        return self.emb.shard(0)

    @property
    def item_table(self) -> torch.Tensor:
        return self.emb.shard(1)

    # ------------------------------------------------------------------ validation loss / retrieval metrics
    def _inputs(self, user_ids, item_ids, category_ids):
        if (category_ids is None) != (self.cat_table is None):
            raise ValueError("category_ids must be given exactly when cfg.n_category_buckets > 0")
        if user_ids.numel() != self.cfg.batch_size or item_ids.numel() != self.cfg.batch_size:
            raise ValueError(f"batch must have {self.cfg.batch_size} pairs per rank (the buffers are sized for it)")
        self.emb.lookup((user_ids, item_ids), self.emb_in)
        if category_ids is not None:
            self.ops.embedding_gather_add_(self.item_tower.acts[0], self.cat_table, category_ids, self.emb.flags[0:1])

    @torch.no_grad()
    def evaluate(self, user_ids: torch.Tensor, item_ids: torch.Tensor, category_ids=None,
                 candidate_sampling_probability=None, sample_weight=None, candidate_ids=None) -> torch.Tensor:
        """Forward only: this rank's validation loss (SUM over its batch; device tensor, unsynchronised).  Collective:
        every rank must call it the same number of times."""
        from .trainer import towers_forward
        cfg, ops, ut, it = self.cfg, self.ops, self.user_tower, self.item_tower
        self._inputs(user_ids, item_ids, category_ids)
        q, c = towers_forward(ut, it) if cfg.symmetric else (ut.forward(), it.forward())
        cp = self._cand_prob(candidate_sampling_probability)
        ci = self._cand_ids(candidate_ids)
        if self.negatives == "local" or not self.collectives:
            return ops.retrieval_fwd(q, c, 1.0 / cfg.temperature, self.ws, self.lse, self.per_row, self.loss, cand_prob=cp,
                                     sample_weight=sample_weight, cand_ids=ci)
        dist.all_gather_into_tensor(self.c_all, c, group=self.group)
        return ops.retrieval_fwd(q, self.c_all, 1.0 / cfg.temperature, self.ws, self.lse, self.per_row, self.loss,
                                 diag_offset=self.rank * cfg.batch_size, cand_prob=cp, sample_weight=sample_weight, cand_ids=ci)

    @torch.no_grad()
    def item_corpus_embeddings(self, item_category_ids: torch.Tensor | None = None) -> torch.Tensor:
        """Item-tower output of EVERY item in global id order ([n_items, scorer_dim], replicated on all ranks): every
        rank runs the tower over its own rows, one all-gather interleaves them (global row = local * world + rank).
        item_category_ids [n_items] (global order) is required iff the model has the category feature."""
        cfg, it, b, w = self.cfg, self.item_tower, self.cfg.batch_size, self.world
        if (item_category_ids is None) != (self.cat_table is None):
            raise ValueError("item_category_ids must be given exactly when cfg.n_category_buckets > 0")
        shard = self.emb.shard(1)
        cap, sd = self.emb.rows_cap[1], it.dims[-1]
        local = torch.zeros(cap, sd, device=self.dev)
        my_cat = None if item_category_ids is None else item_category_ids[self.rank::w].contiguous()
        for s in range(0, shard.shape[0], b):
            e = min(s + b, shard.shape[0])
            it.acts[0][:e - s].copy_(shard[s:e])
            if my_cat is not None:
                self.ops.embedding_gather_add_(it.acts[0][:e - s], self.cat_table, my_cat[s:e], self.emb.flags[0:1])
            it.forward()
            local[s:e].copy_(it.acts[-1][:e - s])
        if not self.collectives:
            return local[:cfg.n_items]
        allr = torch.empty(w * cap, sd, device=self.dev)                  # rank r's rows at [r*cap, (r+1)*cap)
        dist.all_gather_into_tensor(allr, local, group=self.group)
        return allr.view(w, cap, sd).permute(1, 0, 2).reshape(cap * w, sd)[:cfg.n_items].contiguous()

    @torch.no_grad()
    def evaluate_topk(self, user_ids: torch.Tensor, item_ids: torch.Tensor, metric, corpus: torch.Tensor):
        """Updates ``metric`` (metrics.FactorizedTopK) with this rank's (user, true item) pairs ranked against the whole
        corpus (from item_corpus_embeddings()); the user rows come through the exchange, so this is collective."""
        self.emb.lookup((user_ids, item_ids), self.emb_in)
        q = self.user_tower.forward()
        return metric.update_state(q, corpus, item_ids)

    # ------------------------------------------------------------------ checkpoint
    def state_dict(self) -> dict:
        """THIS RANK's part of the checkpoint: its rows of both tables (global rows rank, rank+world, ...), their
        Adagrad accumulators, and the replicated dense parameters (+ category table).  One file per rank; plain
        tensors, so ``torch.load(..., weights_only=True)`` reads it back."""
        # the shards are slices of the combined allocation: torch.save would serialise the WHOLE storage behind a view,
        # so the checkpoint holds compact copies
        own = lambda t: t.detach().contiguous().clone()
        sd = {"config": dict(self.cfg.__dict__), "world": self.world, "rank": self.rank, "negatives": self.negatives,
              "step_index": self.step_index, "dropout_seed": self.dropout_seed,
              "user_shard": own(self.emb.shard(0)), "item_shard": own(self.emb.shard(1)), "dense": own(self.dense_flat)}
        if self.cfg.optimizer == "adagrad":
            sd.update(user_accum=own(self.emb.accum_shard(0)), item_accum=own(self.emb.accum_shard(1)),
                      dense_accum=own(self.dense_accum))
        return sd

    def load_state_dict(self, sd: dict):
        for k in ("n_users", "n_items", "embedding_dim", "tower_dims", "item_tower_dims", "optimizer", "n_category_buckets"):
            if sd["config"].get(k, 0 if k == "n_category_buckets" else None) != getattr(self.cfg, k):
                raise ValueError(f"checkpoint {k}={sd['config'].get(k)!r} does not match the trainer's {getattr(self.cfg, k)!r}")
        if (sd["world"], sd["rank"]) != (self.world, self.rank):
            raise ValueError(f"checkpoint shard is rank {sd['rank']} of {sd['world']}, this is rank {self.rank} of {self.world} "
                             "(row placement is id % world: re-shard through the single-GPU layout to change the world size)")
        self.emb.shard(0).copy_(sd["user_shard"]); self.emb.shard(1).copy_(sd["item_shard"])
        self.dense_flat.copy_(sd["dense"])
        if self.cfg.optimizer == "adagrad":
            self.emb.accum_shard(0).copy_(sd["user_accum"]); self.emb.accum_shard(1).copy_(sd["item_accum"])
            self.dense_accum.copy_(sd["dense_accum"])
        self.step_index = int(sd.get("step_index", 0))
        self.dropout_seed = int(sd.get("dropout_seed", self.dropout_seed))

    def check_ids(self):
        self.emb.check()
