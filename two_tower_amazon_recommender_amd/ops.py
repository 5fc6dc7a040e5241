"""Tensor-level wrappers over the C ABI (``include/twotower_hip.h``).

PyTorch is plumbing here: it owns device memory and the HIP stream; every op below
hands raw device pointers and the current stream to ``libtwotower_hip.so``.  There is
no eager/PyTorch fallback — a missing library or a CPU tensor raises.

Reference anchors: the ops are what ``src/models`` / ``src/training`` of the reference
(docstring stubs, ``src/models/__init__.py:1``) would have run through TensorFlow /
TFRS for the config in ``configs/data_config.yaml:54-71``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import DenseSeg, TT_OPT_ADAGRAD, TT_OPT_SGD

_OPT = {"sgd": TT_OPT_SGD, "adagrad": TT_OPT_ADAGRAD}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, name: str, ndim: int | None = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA/HIP tensor (there is no CPU fallback), got device {t.device}")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    return t


def _p(t):
    return None if t is None else t.data_ptr()


# ----------------------------------------------------------------------------- synthetic
def fill_uniform_(dst: torch.Tensor, seed: int, tensor_id: int, lo: float, scale: float, start: int = 0):
    """dst.flat[i] = fl32(fl32(u(start+i)*scale)+lo); bit-identical to oracle.synth.uniform_f32."""
    _chk(dst, torch.float32, "dst")
    lib = _lib.load()
    _lib.check(lib.tt_fill_uniform_f32(_p(dst), dst.numel(), seed, tensor_id, start, lo, scale, _stream()),
               "tt_fill_uniform_f32")
    return dst


def fill_uniform_rows_(dst: torch.Tensor, seed: int, tensor_id: int, lo: float, scale: float, row_start: int, row_stride: int):
    """dst[lr, :] = row (row_start + lr*row_stride) of the synthetic [*, dim] tensor (a row-sharded table's shard)."""
    _chk(dst, torch.float32, "dst", 2)
    lib = _lib.load()
    _lib.check(lib.tt_fill_uniform_rows_f32(_p(dst), dst.shape[0], dst.shape[1], row_start, row_stride, seed, tensor_id,
                                            lo, scale, _stream()), "tt_fill_uniform_rows_f32")
    return dst


def fill_ids_(dst: torch.Tensor, seed: int, tensor_id: int, num_rows: int, variant: str = "U", start: int = 0):
    _chk(dst, torch.int64, "dst")
    v = {"U": _lib.TT_IDS_UNIFORM, "Z": _lib.TT_IDS_POWERLAW}[variant]
    lib = _lib.load()
    _lib.check(lib.tt_fill_ids_i64(_p(dst), dst.numel(), seed, tensor_id, start, num_rows, v, _stream()),
               "tt_fill_ids_i64")
    return dst


# ----------------------------------------------------------------------------- a6/a7 id encoding
def strings_to_padded_bytes(values) -> "torch.Tensor":
    """Host helper: list of str -> zero-padded [n, width] uint8 matrix of UTF-8 bytes (width a multiple of 8)."""
    import numpy as np
    enc = [v.encode("utf-8") for v in values]
    width = max(8, (max((len(b) for b in enc), default=1) + 7) // 8 * 8)
    mat = np.zeros((len(enc), width), dtype=np.uint8)
    for i, b in enumerate(enc):
        if b"\0" in b:
            raise ValueError("id strings must not contain NUL bytes")
        mat[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    return torch.from_numpy(mat)


def encode_ids(rows_u8: torch.Tensor):
    """codes[i] = rank of row i among the sorted distinct rows (the reference's sorted-enumerate / LabelEncoder ids).
    Returns (int64 codes [n], int32 n_unique [1]) on the device."""
    _chk(rows_u8, torch.uint8, "rows_u8", 2)
    n, width = rows_u8.shape
    lib = _lib.load()
    ws = torch.empty(int(lib.tt_encode_ids_workspace_bytes(n)), dtype=torch.uint8, device=rows_u8.device)
    codes = torch.empty(n, dtype=torch.int64, device=rows_u8.device)
    nuniq = torch.zeros(1, dtype=torch.int32, device=rows_u8.device)
    _lib.check(lib.tt_encode_ids_u8(_p(rows_u8), n, width, _p(ws), ws.numel(), _p(codes), _p(nuniq), _stream()),
               "tt_encode_ids_u8")
    return codes, nuniq


# ----------------------------------------------------------------------------- a1 gather
def embedding_gather(table: torch.Tensor, ids: torch.Tensor, out: torch.Tensor | None = None,
                     oob_flag: torch.Tensor | None = None) -> torch.Tensor:
    """out[b,:] = table[ids[b],:].  ``oob_flag`` (int32[1]) is set to 1 on any out-of-range id."""
    _chk(table, torch.float32, "table", 2)
    _chk(ids, torch.int64, "ids", 1)
    n, d = ids.numel(), table.shape[1]
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=table.device)
    _chk(out, torch.float32, "out", 2)
    if tuple(out.shape) != (n, d):
        raise RuntimeError(f"embedding_gather: out must be [{n}, {d}] (n_ids, dim), got {tuple(out.shape)}")
    if oob_flag is not None:
        _chk(oob_flag, torch.int32, "oob_flag")
    lib = _lib.load()
    _lib.check(lib.tt_embedding_gather_f32(_p(table), table.shape[0], d, _p(ids), n, _p(out), _p(oob_flag), _stream()),
               "tt_embedding_gather_f32")
    return out


def embedding_gather2(table_a, ids_a, out_a, table_b, ids_b, out_b, oob_flag=None):
    """Both towers' lookups in one launch."""
    for t, nme in ((table_a, "table_a"), (table_b, "table_b"), (out_a, "out_a"), (out_b, "out_b")):
        _chk(t, torch.float32, nme, 2)
    _chk(ids_a, torch.int64, "ids_a", 1)
    _chk(ids_b, torch.int64, "ids_b", 1)
    if ids_a.numel() != ids_b.numel() or table_a.shape[1] != table_b.shape[1]:
        raise RuntimeError("embedding_gather2: both lookups must share n_ids and dim")
    want = (ids_a.numel(), table_a.shape[1])
    if tuple(out_a.shape) != want or tuple(out_b.shape) != want:
        raise RuntimeError(f"embedding_gather2: outputs must be [{want[0]}, {want[1]}] (n_ids, dim), got "
                           f"{tuple(out_a.shape)} and {tuple(out_b.shape)}")
    lib = _lib.load()
    _lib.check(lib.tt_embedding_gather2_f32(_p(table_a), table_a.shape[0], _p(ids_a), _p(out_a),
                                            _p(table_b), table_b.shape[0], _p(ids_b), _p(out_b),
                                            table_a.shape[1], ids_a.numel(), _p(oob_flag), _stream()),
               "tt_embedding_gather2_f32")
    return out_a, out_b


def embedding_gather_add_(out, table, ids, oob_flag=None):
    """out[p, :] += table[ids[p], :] — a further feature summed into a tower input (hashed category, cfg5)."""
    _chk(table, torch.float32, "table", 2)
    _chk(ids, torch.int64, "ids", 1)
    _chk(out, torch.float32, "out", 2)
    if out.shape[0] != ids.numel() or out.shape[1] != table.shape[1]:
        raise RuntimeError("embedding_gather_add_: out must be [n_ids, dim]")
    lib = _lib.load()
    _lib.check(lib.tt_embedding_gather_add_f32(_p(table), table.shape[0], table.shape[1], _p(ids), ids.numel(), _p(out),
                                               _p(oob_flag), _stream()), "tt_embedding_gather_add_f32")
    return out


def hash_buckets(rows_u8: torch.Tensor, n_buckets: int) -> torch.Tensor:
    """int64 bucket of every zero-padded byte row: FNV-1a-64 mod n_buckets (oracle/hashing.py)."""
    _chk(rows_u8, torch.uint8, "rows_u8", 2)
    n, width = rows_u8.shape
    out = torch.empty(n, dtype=torch.int64, device=rows_u8.device)
    lib = _lib.load()
    _lib.check(lib.tt_hash_bucket_u8(_p(rows_u8), n, width, n_buckets, _p(out), _stream()), "tt_hash_bucket_u8")
    return out


# ----------------------------------------------------------------------------- sharded routing
def route_by_owner(ids, world: int, num_rows: int, cap: int, send_ids, pos_flat, flags=None):
    _chk(ids, torch.int64, "ids", 1)
    _chk(send_ids, torch.int64, "send_ids")
    _chk(pos_flat, torch.int64, "pos_flat")
    if send_ids.numel() < world * cap or pos_flat.numel() < ids.numel():
        raise RuntimeError("route_by_owner: output buffers too small")
    lib = _lib.load()
    _lib.check(lib.tt_route_by_owner_i64(_p(ids), ids.numel(), world, num_rows, cap, _p(send_ids), _p(pos_flat), _p(flags),
                                         _stream()), "tt_route_by_owner_i64")


def route_tables_by_owner(ids_list, world: int, num_rows_list, local_offsets, cap: int, send_ids, pos_flats, flags=None):
    """Several tables, one launch: send_ids is [world][n_tables][cap] (see tt_route_tables_by_owner_i64)."""
    t = len(ids_list)
    n = ids_list[0].numel()
    for ids, pos in zip(ids_list, pos_flats):
        _chk(ids, torch.int64, "ids", 1)
        _chk(pos, torch.int64, "pos_flat")
        if ids.numel() != n or pos.numel() < n:
            raise RuntimeError("route_tables_by_owner: every table needs n_ids ids and a pos_flat of n_ids")
    _chk(send_ids, torch.int64, "send_ids")
    if send_ids.numel() < world * t * cap:
        raise RuntimeError("route_tables_by_owner: send_ids too small")
    arr = (_lib.RouteTable * t)(*[_lib.RouteTable(_p(ids_list[i]), int(num_rows_list[i]), int(local_offsets[i]), _p(pos_flats[i]))
                                  for i in range(t)])
    lib = _lib.load()
    _lib.check(lib.tt_route_tables_by_owner_i64(arr, t, n, world, cap, _p(send_ids), _p(flags), _stream()),
               "tt_route_tables_by_owner_i64")


def scatter_rows(src, idx, dst):
    _chk(src, torch.float32, "src", 2)
    _chk(idx, torch.int64, "idx", 1)
    _chk(dst, torch.float32, "dst", 2)
    if src.shape[1] != dst.shape[1] or idx.numel() != src.shape[0]:
        raise RuntimeError("scatter_rows: shape mismatch")
    lib = _lib.load()
    _lib.check(lib.tt_scatter_rows_f32(_p(src), _p(idx), src.shape[0], src.shape[1], _p(dst), dst.shape[0], _stream()),
               "tt_scatter_rows_f32")


# ----------------------------------------------------------------------------- a5 sparse optimizer
class SparsePlan:
    """Sorted (id, position) list of one id batch; reusable buffers."""

    def __init__(self, n_ids: int, device):
        lib = _lib.load()
        self.n_ids = n_ids
        self.ws_bytes = int(lib.tt_sparse_plan_workspace_bytes(n_ids))
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.sorted_ids = torch.empty(n_ids, dtype=torch.int64, device=device)
        self.order = torch.empty(n_ids, dtype=torch.int32, device=device)
        self._apply_ws = None

    def apply_ws(self, dim: int) -> torch.Tensor:
        """Piece-sum workspace of the apply kernels for rows of ``dim`` floats (allocated once)."""
        need = int(_lib.load().tt_sparse_apply_workspace_bytes(self.n_ids, dim))
        if self._apply_ws is None or self._apply_ws.numel() < need:
            self._apply_ws = torch.zeros(need, dtype=torch.uint8, device=self.sorted_ids.device)   # contract: zeroed once
        return self._apply_ws

    def run(self, ids: torch.Tensor, num_rows: int) -> "SparsePlan":
        _chk(ids, torch.int64, "ids", 1)
        if ids.numel() != self.n_ids:
            raise RuntimeError(f"SparsePlan: built for {self.n_ids} ids, got {ids.numel()}")
        lib = _lib.load()
        _lib.check(lib.tt_sparse_plan(_p(ids), self.n_ids, num_rows, _p(self.workspace), self.ws_bytes,
                                      _p(self.sorted_ids), _p(self.order), _stream()), "tt_sparse_plan")
        return self


def sparse_plan_batched(plans, ids_list, num_rows_list):
    """Sort up to 4 id lists (user, item, hashed category, ...) in ONE launch (key-range partitions: csrc/sort.hip)."""
    n = len(plans)
    args = []
    for plan, ids, rows in zip(plans, ids_list, num_rows_list):
        _chk(ids, torch.int64, "ids", 1)
        if ids.numel() != plan.n_ids:
            raise RuntimeError(f"SparsePlan: built for {plan.n_ids} ids, got {ids.numel()}")
        args.append(_lib.SparsePlanArgs(_p(ids), plan.n_ids, int(rows), _p(plan.workspace), plan.ws_bytes,
                                        _p(plan.sorted_ids), _p(plan.order)))
    arr = (_lib.SparsePlanArgs * n)(*args)
    _lib.check(_lib.load().tt_sparse_plan_batched(arr, n, _stream()), "tt_sparse_plan_batched")


def sparse_sgd_(table, grads, plan: SparsePlan, lr: float):
    _chk(table, torch.float32, "table", 2)
    _chk(grads, torch.float32, "grads", 2)
    lib = _lib.load()
    _lib.check(lib.tt_sparse_sgd_f32(_p(table), table.shape[0], table.shape[1], _p(grads), _p(plan.sorted_ids),
                                     _p(plan.order), plan.n_ids, lr, _p(plan.apply_ws(table.shape[1])), _stream()),
               "tt_sparse_sgd_f32")
    return table


def sparse_adagrad_(table, accum, grads, plan: SparsePlan, lr: float, eps: float = 1e-7):
    _chk(table, torch.float32, "table", 2)
    _chk(accum, torch.float32, "accum", 2)
    _chk(grads, torch.float32, "grads", 2)
    lib = _lib.load()
    _lib.check(lib.tt_sparse_adagrad_f32(_p(table), _p(accum), table.shape[0], table.shape[1], _p(grads),
                                         _p(plan.sorted_ids), _p(plan.order), plan.n_ids, lr, eps,
                                         _p(plan.apply_ws(table.shape[1])), _stream()), "tt_sparse_adagrad_f32")
    return table


def sparse_update2_(opt: str, table_a, accum_a, grads_a, plan_a: SparsePlan,
                    table_b, accum_b, grads_b, plan_b: SparsePlan, lr: float, eps: float = 1e-7):
    """User and item table updates in one launch."""
    lib = _lib.load()
    _lib.check(lib.tt_sparse_update2_f32(_OPT[opt], _p(table_a), _p(accum_a), table_a.shape[0], _p(grads_a),
                                         _p(plan_a.sorted_ids), _p(plan_a.order),
                                         _p(table_b), _p(accum_b), table_b.shape[0], _p(grads_b),
                                         _p(plan_b.sorted_ids), _p(plan_b.order),
                                         table_a.shape[1], plan_a.n_ids, lr, eps, _p(plan_a.apply_ws(table_a.shape[1])),
                                         _p(plan_b.apply_ws(table_b.shape[1])), _stream()), "tt_sparse_update2_f32")


# ----------------------------------------------------------------------------- a2 dense layers
MAX_FUSED_LOOKUP_ROWS = 32768       # tt_dense_lookup: the ids of one dW split are staged in LDS


class IdBuckets:
    """Row-range id lists (``tt_id_buckets``, ABI v9) for one train step's tables: the forward lookup (``make_lookup(...,
    buckets=b.desc(t, gen))``) appends every id to the list of the row range the optimizer launch's sorting workgroup owns, and
    ``optimizer_step_ids_(..., buckets=[...])`` of the same step reads ~64 entries per workgroup instead of all the ids.
    ``cap == 0``: the shape takes no lists (dim > 128 or more than 16384 ids)."""

    def __init__(self, table_rows, dim: int, n_ids: int, segs, device):
        lib = _lib.load()
        n = len(table_rows)
        rows = (_lib.C.c_int64 * n)(*table_rows)
        groups, width, cap = (_lib.C.c_int32 * n)(), (_lib.C.c_uint32 * n)(), _lib.C.c_int32(0)
        arr_s = (DenseSeg * len(segs))(*segs)
        _lib.check(lib.tt_optimizer_ids_geometry(rows, n, dim, n_ids, arr_s, len(segs), groups, width, _lib.C.byref(cap)),
                   "tt_optimizer_ids_geometry")
        self.groups, self.width, self.cap = list(groups), list(width), int(cap.value)
        self.per = int(lib.tt_id_buckets_workspace_bytes())
        self.ws = torch.zeros(n * self.per, dtype=torch.uint8, device=device)

    def desc(self, t: int, gen: int) -> "_lib.IdBuckets":
        base = self.ws.data_ptr() + t * self.per
        return _lib.IdBuckets(base, base + self.COUNT_BYTES, self.groups[t], self.width[t], self.cap, gen & 0xFFFFFFFF)

    COUNT_BYTES = 256 * 256          # one counter per 256-byte line (csrc/common.h: kBucketGroupsMax * kBucketCountStride * 4)

    def counts(self, t: int) -> torch.Tensor:
        return self.ws[t * self.per: t * self.per + self.COUNT_BYTES].view(torch.int32)[:: 64][: self.groups[t]]


def make_lookup(table, ids, table2=None, ids2=None, oob_flag=None, buckets=None) -> "_lib.DenseLookup":
    """The embedding lookup fused into a tower's FIRST Dense layer (``tt_dense_lookup``): the layer's input row r is
    ``table[ids[r]]`` (+ ``table2[ids2[r]]``), read straight into the GEMM tiles — never written to HBM."""
    _chk(table, torch.float32, "lookup table", 2)
    _chk(ids, torch.int64, "lookup ids", 1)
    if (table2 is None) != (ids2 is None):
        raise RuntimeError("make_lookup: table2 and ids2 go together")
    if table2 is not None:
        _chk(table2, torch.float32, "lookup table2", 2)
        _chk(ids2, torch.int64, "lookup ids2", 1)
        if table2.shape[1] != table.shape[1] or ids2.numel() != ids.numel():
            raise RuntimeError("make_lookup: table2 / ids2 must match table's width and the number of ids")
    if oob_flag is not None:
        _chk(oob_flag, torch.int32, "oob_flag")
    if ids.numel() > MAX_FUSED_LOOKUP_ROWS:
        raise RuntimeError(f"make_lookup: at most {MAX_FUSED_LOOKUP_ROWS} rows per fused lookup")
    lk = _lib.DenseLookup(_p(table), _p(ids), table.shape[0], _p(table2), _p(ids2), 0 if table2 is None else table2.shape[0],
                          _p(oob_flag))
    if buckets is not None:
        lk.buckets = buckets
    lk._keep = (table, ids, table2, ids2, oob_flag)      # the struct holds raw pointers: keep the tensors alive with it
    lk._mk = (ids.numel(), table.shape[1])
    return lk


def _no_lookup():
    return _lib.DenseLookup()


def _in_shape(x, lookup):
    if lookup is not None:
        return lookup._mk
    _chk(x, torch.float32, "x", 2)
    return x.shape[0], x.shape[1]


def relu_bits_like(m: int, n: int, device) -> torch.Tensor:
    """[m, n/32] int32 buffer for the sign bits a forward layer writes beside its output (n % 32 == 0)."""
    if n % 32:
        raise RuntimeError(f"relu bits need a layer width that is a multiple of 32, got {n}")
    return torch.empty((m, n // 32), dtype=torch.int32, device=device)


def _chk_bits(bits, m, n, what):
    if bits is None:
        return
    if bits.dtype != torch.int32 or not bits.is_cuda or not bits.is_contiguous() or tuple(bits.shape) != (m, n // 32) or n % 32:
        raise RuntimeError(f"{what}: expected a contiguous CUDA int32 tensor [{m}, {n}/32] (width a multiple of 32), got "
                           f"{bits.dtype} {tuple(bits.shape)}")


def dense_fwd(x, w, b, relu: bool, out=None, dropout=None, lookup=None, relu_bits=None):
    """y = act(x@w+b); ``dropout`` = (rate, seed, tensor_id, counter_offset) applies inverted dropout to y.
    With ``lookup`` (make_lookup) x is ignored: the input rows come from the embedding table.
    ``relu_bits`` (relu_bits_like): also written, bit = (y > 0) — the next layer's backward takes it as its dx mask."""
    _chk(w, torch.float32, "w", 2)
    if b is not None:
        _chk(b, torch.float32, "b", 1)
    m, k = _in_shape(x, lookup)
    n = w.shape[1]
    if w.shape[0] != k:
        raise RuntimeError(f"dense_fwd: input is [{m},{k}] but w is {tuple(w.shape)}")
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=w.device)
    _chk(out, torch.float32, "out", 2)
    if tuple(out.shape) != (m, n):
        raise RuntimeError(f"dense_fwd: out must be [{m},{n}], got {tuple(out.shape)}")
    rate, seed, tid, off = dropout if dropout is not None else (0.0, 0, 0, 0)
    _chk_bits(relu_bits, m, n, "dense_fwd: relu_bits")
    arr = (_lib.DenseFwdArgs * 1)(_lib.DenseFwdArgs(None if lookup is not None else _p(x), _p(w), _p(b), _p(out), tid,
                                                    lookup if lookup is not None else _no_lookup(), _p(relu_bits)))
    _lib.check(_lib.load().tt_dense_fwd_batched_f32(arr, 1, m, k, n, int(relu), rate, seed, off, _stream()),
               "tt_dense_fwd_batched_f32")
    return out


def dense_bwd_num_slabs(m: int) -> int:
    return int(_lib.load().tt_dense_bwd_num_slabs(m))


def dense_bwd(x, w, dz, dx, dx_relu_src, dw_slabs, db_slabs, dx_scale: float = 1.0, lookup=None, dx_relu_bits=None):
    """dx = dz@w^T (* (dx_relu_src>0)); dw_slabs/db_slabs get the split-K partials.  With ``lookup`` the layer's input
    (needed by dw = x^T dz) is read from the embedding table.  ``dx_relu_bits`` (the sign bits the previous layer's
    forward wrote) replaces ``dx_relu_src`` as the mask."""
    _chk(w, torch.float32, "w", 2)
    _chk(dz, torch.float32, "dz", 2)
    m, k = _in_shape(x, lookup)
    n = w.shape[1]
    if dx is not None:
        _chk(dx, torch.float32, "dx", 2)
    if dx_relu_src is not None:
        _chk(dx_relu_src, torch.float32, "dx_relu_src", 2)
    ns = dense_bwd_num_slabs(m)
    if dw_slabs is not None or db_slabs is not None:      # both None: dx only
        _chk(dw_slabs, torch.float32, "dw_slabs")
        _chk(db_slabs, torch.float32, "db_slabs")
        if dw_slabs.numel() < ns * k * n or db_slabs.numel() < ns * n:
            raise RuntimeError("dense_bwd: slab buffers too small")
    _chk_bits(dx_relu_bits, m, k, "dense_bwd: dx_relu_bits")
    arr = (_lib.DenseBwdArgs * 1)(_lib.DenseBwdArgs(None if lookup is not None else _p(x), _p(w), _p(dz), _p(dx), _p(dx_relu_src),
                                                    _p(dw_slabs), _p(db_slabs), lookup if lookup is not None else _no_lookup(),
                                                    _p(dx_relu_bits)))
    _lib.check(_lib.load().tt_dense_bwd_batched_f32(arr, 1, dx_scale, m, k, n, _stream()), "tt_dense_bwd_batched_f32")
    return ns


def dense_fwd2(xs, ws, bs, ys, relu: bool, dropout=None, lookups=None, relu_bits=(None, None)):
    """Layer l of both towers in one launch: ys[i] = act(xs[i] @ ws[i] + bs[i]).  dropout = (rate, seed, (tid_a, tid_b), offset).
    lookups = (lookup_a, lookup_b): the towers' first layer reads its input rows from the embedding tables."""
    m, k = _in_shape(xs[0], None if lookups is None else lookups[0])
    n = ws[0].shape[1]
    rate, seed, tids, off = dropout if dropout is not None else (0.0, 0, (0, 0), 0)
    for i in range(2):
        _chk_bits(relu_bits[i], m, n, "dense_fwd2: relu_bits")
    arr = (_lib.DenseFwdArgs * 2)(*[_lib.DenseFwdArgs(None if lookups is not None else _p(xs[i]), _p(ws[i]), _p(bs[i]), _p(ys[i]),
                                                      tids[i], lookups[i] if lookups is not None else _no_lookup(),
                                                      _p(relu_bits[i])) for i in range(2)])
    _lib.check(_lib.load().tt_dense_fwd_batched_f32(arr, 2, m, k, n, int(relu), rate, seed, off, _stream()),
               "tt_dense_fwd_batched_f32")


def tower_fwd2_supported(m: int, k0: int, h: int, n1: int) -> bool:
    """Shapes the fused two-layer tower forward takes (csrc/tower.hip).  TT_FUSED_TOWER=0 switches it off (A/B)."""
    import os
    if os.environ.get("TT_FUSED_TOWER", "1") == "0":
        return False
    return bool(_lib.load().tt_tower_fwd2_supported(m, k0, h, n1))


def tower_fwd2(xs, w0s, b0s, hs, h_bits, w1s, b1s, ys, dropout=None, lookups=None):
    """h = relu(x @ w0 + b0) [dropout], y = h @ w1 + b1 for both towers in ONE launch (the hidden tile stays in LDS; h and
    its sign bits are still written for the backward pass).  Bit-identical to dense_fwd2 called for each layer.
    dropout = (rate, seed, (tid_a, tid_b), offset) for the hidden layer; lookups as in dense_fwd2."""
    m, k0 = _in_shape(xs[0], None if lookups is None else lookups[0])
    h, n1 = w0s[0].shape[1], w1s[0].shape[1]
    rate, seed, tids, off = dropout if dropout is not None else (0.0, 0, (0, 0), 0)
    for i in range(2):
        _chk(w0s[i], torch.float32, "w0", 2); _chk(w1s[i], torch.float32, "w1", 2)
        _chk(hs[i], torch.float32, "h", 2); _chk(ys[i], torch.float32, "y", 2)
        if tuple(w0s[i].shape) != (k0, h) or tuple(w1s[i].shape) != (h, n1) or tuple(hs[i].shape) != (m, h) or tuple(ys[i].shape) != (m, n1):
            raise RuntimeError("tower_fwd2: shape mismatch between the layers' weights and buffers")
        _chk_bits(h_bits[i], m, h, "tower_fwd2: h_bits")
    l0 = (_lib.DenseFwdArgs * 2)(*[_lib.DenseFwdArgs(None if lookups is not None else _p(xs[i]), _p(w0s[i]), _p(b0s[i]), _p(hs[i]),
                                                     tids[i], lookups[i] if lookups is not None else _no_lookup(), _p(h_bits[i]))
                                   for i in range(2)])
    l1 = (_lib.DenseFwdArgs * 2)(*[_lib.DenseFwdArgs(_p(hs[i]), _p(w1s[i]), _p(b1s[i]), _p(ys[i]), 0, _no_lookup(), None) for i in range(2)])
    _lib.check(_lib.load().tt_tower_fwd2_batched_f32(l0, l1, 2, m, k0, h, n1, rate, seed, off, _stream()), "tt_tower_fwd2_batched_f32")


def dense_bwd2(xs, ws, dzs, dxs, dx_relu_srcs, dw_slabs, db_slabs, dx_scale: float = 1.0, lookups=None, dx_relu_bits=(None, None),
               riders=None):
    """Backward of layer l of both towers: one launch (dx and dw+db tiles side by side; dx only / dw only: one each).
    ``riders = (segs, opt, lr, eps)``: the dense update of ANOTHER layer's segments in the same launch
    (``tt_dense_bwd_batched_update_f32``; bit-identical to ``dense_update_`` behind the launch)."""
    m, k = _in_shape(xs[0], None if lookups is None else lookups[0])
    n = ws[0].shape[1]
    for i in range(2):
        _chk_bits(dx_relu_bits[i], m, k, "dense_bwd2: dx_relu_bits")
    arr = (_lib.DenseBwdArgs * 2)(*[_lib.DenseBwdArgs(None if lookups is not None else _p(xs[i]), _p(ws[i]), _p(dzs[i]), _p(dxs[i]),
                                                     _p(dx_relu_srcs[i]), _p(dw_slabs[i]), _p(db_slabs[i]),
                                                     lookups[i] if lookups is not None else _no_lookup(),
                                                     _p(dx_relu_bits[i])) for i in range(2)])
    if riders is not None:
        segs, opt, lr, eps = riders
        arr_s = (DenseSeg * len(segs))(*segs)
        _lib.check(_lib.load().tt_dense_bwd_batched_update_f32(arr, 2, dx_scale, m, k, n, arr_s, len(segs), _OPT[opt], lr, eps, _stream()),
                   "tt_dense_bwd_batched_update_f32")
        return
    _lib.check(_lib.load().tt_dense_bwd_batched_f32(arr, 2, dx_scale, m, k, n, _stream()), "tt_dense_bwd_batched_f32")


def _bwd_args(xs, ws, dzs, dxs, dx_relu_srcs, dw_slabs, db_slabs, lookups, dx_relu_bits):
    return (_lib.DenseBwdArgs * 2)(*[_lib.DenseBwdArgs(None if lookups is not None else _p(xs[i]), _p(ws[i]), _p(dzs[i]), _p(dxs[i]),
                                                      _p(dx_relu_srcs[i]), _p(dw_slabs[i]), _p(db_slabs[i]),
                                                      lookups[i] if lookups is not None else _no_lookup(),
                                                      _p(dx_relu_bits[i])) for i in range(2)])


def tower_bwd2_supported(m: int, k0: int, k1: int, n: int) -> bool:
    return bool(_lib.load().tt_tower_bwd2_supported(m, k0, k1, n))


def tower_bwd2_workspace(m: int, device) -> torch.Tensor:
    """The dependency counters of ``tower_bwd2`` (zeroed once; every launch leaves them zeroed; int32 word 4*(m//64) = error)."""
    return torch.zeros(int(_lib.load().tt_tower_bwd2_workspace_bytes(m)), dtype=torch.uint8, device=device)


def tower_bwd2(upper: dict, lower: dict, workspace: torch.Tensor, dx_scale_upper: float = 1.0, dx_scale_lower: float = 1.0):
    """Backward of layers l (``upper``) and l-1 (``lower``) of both towers in ONE launch (``tt_tower_bwd2_batched_f32``); each
    dict holds ``dense_bwd2``'s arguments: xs, ws, dzs, dxs, dx_relu_srcs, dw_slabs, db_slabs and optionally lookups, dx_relu_bits.
    ``upper['dxs'][i]`` must BE ``lower['dzs'][i]``.  Bit-identical to the two ``dense_bwd2`` calls."""
    none2 = (None, None)
    lk = lower.get("lookups")
    m, k0 = _in_shape(lower["xs"][0], None if lk is None else lk[0])
    k1, n = upper["ws"][0].shape[0], upper["ws"][0].shape[1]
    bu, bl = upper.get("dx_relu_bits", none2), lower.get("dx_relu_bits", none2)
    for i in range(2):
        _chk_bits(bu[i], m, k1, "tower_bwd2: upper dx_relu_bits")
        _chk_bits(bl[i], m, k0, "tower_bwd2: lower dx_relu_bits")
    au = _bwd_args(upper["xs"], upper["ws"], upper["dzs"], upper["dxs"], upper["dx_relu_srcs"], upper["dw_slabs"], upper["db_slabs"], None, bu)
    al = _bwd_args(lower["xs"], lower["ws"], lower["dzs"], lower["dxs"], lower["dx_relu_srcs"], lower["dw_slabs"], lower["db_slabs"], lk, bl)
    _lib.check(_lib.load().tt_tower_bwd2_batched_f32(au, al, 2, dx_scale_upper, dx_scale_lower, m, k0, k1, n, _p(workspace), _stream()),
               "tt_tower_bwd2_batched_f32")


def dense_update_(segs: list[DenseSeg], opt: str, lr: float, eps: float = 1e-7, apply: bool = True):
    arr = (DenseSeg * len(segs))(*segs)
    lib = _lib.load()
    _lib.check(lib.tt_dense_update_f32(arr, len(segs), _OPT[opt], int(apply), lr, eps, _stream()), "tt_dense_update_f32")


def optimizer_step_(opt: str, tables, segs: list[DenseSeg], lr: float, eps: float = 1e-7):
    """The train step's whole optimizer in ONE launch: ``tables`` = [(table, accum or None, grads, plan), ...] (up to 3
    embedding tables of one width whose plans hold the same number of ids) + the dense tower segments."""
    dim, n_ids = tables[0][0].shape[1], tables[0][3].n_ids
    arr_t = (_lib.SparseTable * len(tables))()
    for i, (table, accum, grads, plan) in enumerate(tables):
        _chk(table, torch.float32, "table", 2)
        _chk(grads, torch.float32, "grads", 2)
        if accum is not None:
            _chk(accum, torch.float32, "accum", 2)
        if table.shape[1] != dim or plan.n_ids != n_ids or tuple(grads.shape) != (n_ids, dim):
            raise RuntimeError("optimizer_step_: every table needs the same dim, the same number of ids and [n_ids, dim] gradients")
        arr_t[i] = _lib.SparseTable(_p(table), _p(accum), table.shape[0], _p(grads), _p(plan.sorted_ids), _p(plan.order),
                                    _p(plan.apply_ws(dim)))
    arr_s = (DenseSeg * len(segs))(*segs)
    _lib.check(_lib.load().tt_optimizer_step_f32(_OPT[opt], arr_t, len(tables), dim, n_ids, arr_s, len(segs), lr, eps, _stream()),
               "tt_optimizer_step_f32")


def sparse_plan_max_lds_ids() -> int:
    """Longest id list the one-launch LDS sort takes: 16384."""
    return int(_lib.load().tt_sparse_plan_max_lds_ids())


def optimizer_ids_max_ids() -> int:
    """Longest id list per table the optimizer step from raw ids takes: 65536 (beyond 16384: the long-list kernel)."""
    return int(_lib.load().tt_optimizer_ids_max_ids())


def id_range_load_(out_max: torch.Tensor, ids_list, table_rows, dim: int, segs: list[DenseSeg]):
    """Skew probe (``tt_id_range_load``): out_max[t] (int32, device) = the most ids of ``ids_list[t]`` in one of table t's row
    ranges - what ONE workgroup of the one-launch optimizer would have to sort and apply.  One small launch; the caller copies
    ``out_max`` to pinned host memory without waiting (TwoTowerTrainer.poll_ids)."""
    n = len(ids_list)
    _chk(out_max, torch.int32, "out_max", 1)
    if out_max.numel() < n:
        raise RuntimeError("id_range_load_: out_max needs one int32 per table")
    for ids in ids_list:
        _chk(ids, torch.int64, "ids", 1)
        if ids.numel() != ids_list[0].numel():
            raise RuntimeError("id_range_load_: every table needs the same number of ids")
    ptrs = (_lib.C.c_void_p * n)(*[ids.data_ptr() for ids in ids_list])
    rows = (_lib.C.c_int64 * n)(*[int(r) for r in table_rows])
    arr_s = (DenseSeg * len(segs))(*segs)
    _lib.check(_lib.load().tt_id_range_load(ptrs, rows, n, dim, ids_list[0].numel(), arr_s, len(segs), _p(out_max), _stream()),
               "tt_id_range_load")
    return out_max


def optimizer_step_ids_(opt: str, tables, segs: list[DenseSeg], lr: float, eps: float = 1e-7, buckets=None):
    """The same step from the RAW ids, no sort-plan launch: ``tables`` = [(table, accum or None, grads, ids, plan), ...]
    (``plan`` only lends its apply workspace); n_ids <= optimizer_ids_max_ids().  Bit-identical to
    ``sparse_plan_batched`` + ``optimizer_step_``."""
    dim, n_ids = tables[0][0].shape[1], tables[0][3].numel()
    arr_t = (_lib.SparseTableIds * len(tables))()
    for i, (table, accum, grads, ids, plan) in enumerate(tables):
        _chk(table, torch.float32, "table", 2)
        _chk(grads, torch.float32, "grads", 2)
        _chk(ids, torch.int64, "ids", 1)
        if accum is not None:
            _chk(accum, torch.float32, "accum", 2)
            if accum.shape != table.shape:
                raise RuntimeError("optimizer_step_ids_: accum must have the table's shape")
        if table.shape[1] != dim or ids.numel() != n_ids or plan.n_ids != n_ids or tuple(grads.shape) != (n_ids, dim):
            raise RuntimeError("optimizer_step_ids_: every table needs the same dim, the same number of ids and [n_ids, dim] gradients")
        arr_t[i] = _lib.SparseTableIds(_p(table), _p(accum), table.shape[0], _p(grads), _p(ids), _p(plan.apply_ws(dim)))
        if buckets is not None and buckets[i] is not None:       # the row-range lists this step's forward lookup filled
            arr_t[i].buckets = buckets[i]
    arr_s = (DenseSeg * len(segs))(*segs)
    _lib.check(_lib.load().tt_optimizer_step_ids_f32(_OPT[opt], arr_t, len(tables), dim, n_ids, arr_s, len(segs), lr, eps, _stream()),
               "tt_optimizer_step_ids_f32")


def make_dense_seg(param, accum, grad_slabs, n_slabs: int, l2: float, grad_out=None) -> DenseSeg:
    count = param.numel()
    return DenseSeg(_p(param), _p(accum), _p(grad_slabs), _p(grad_out), count, count, n_slabs, l2)


# ----------------------------------------------------------------------------- a3+a4 retrieval
SCORER_PRECISIONS = ("f32", "bf16x3")


def retrieval_workspace_bytes(nq: int, nc: int, dim: int) -> int:
    """Workspace of the fused training entries (retrieval_fwd_bwd): includes the [nq, nc] f32 logit buffer pass 2 reads back."""
    return int(_lib.load().tt_retrieval_workspace_bytes(nq, nc, dim))


def retrieval_fwd_workspace_bytes(nq: int, nc: int, dim: int) -> int:
    """Workspace of the forward-only / separate-backward entries (no logit buffer)."""
    return int(_lib.load().tt_retrieval_fwd_workspace_bytes(nq, nc, dim))


def _chk_retrieval(q, c, sample_weight=None, cand_prob=None, cand_ids=None, hard_thr=None, lse=None, per_row=None,
                   dq=None, dc=None):
    """dtype / device / shape checks shared by the retrieval entry points: the kernels index the optional vectors by
    query (nq) or candidate (nc) without looking at their length."""
    _chk(q, torch.float32, "query_embeddings", 2)
    _chk(c, torch.float32, "candidate_embeddings", 2)
    if q.shape[1] != c.shape[1]:
        raise RuntimeError(f"retrieval: embedding dims differ: {q.shape[1]} vs {c.shape[1]}")
    nq, nc = q.shape[0], c.shape[0]
    for t, dt, name, n in ((sample_weight, torch.float32, "sample_weight", nq), (hard_thr, torch.float32, "hard_thr", nq),
                           (lse, torch.float32, "lse", nq), (per_row, torch.float32, "per_row", nq),
                           (cand_prob, torch.float32, "candidate_sampling_probability", nc),
                           (cand_ids, torch.int64, "candidate_ids", nc)):
        if t is not None:
            _chk(t, dt, name, 1)
            if t.numel() != n:
                raise RuntimeError(f"retrieval: {name} must have {n} entries, got {t.numel()}")
    for t, name, ref in ((dq, "dq", q), (dc, "dc", c)):
        if t is not None:
            _chk(t, torch.float32, name, 2)
            if tuple(t.shape) != tuple(ref.shape):
                raise RuntimeError(f"retrieval: {name} must be {tuple(ref.shape)}, got {tuple(t.shape)}")


def retrieval_rank_workspace_bytes(nq: int, nc: int, dim: int) -> int:
    """Workspace of the metric (rank) pass alone: no gradient slabs (those make the full workspace as large as the
    candidate corpus when nc >= 65536)."""
    return int(_lib.load().tt_retrieval_rank_workspace_bytes(nq, nc, dim))


def retrieval_fwd(q, c, inv_temperature: float, workspace, lse, per_row, loss, sample_weight=None,
                  cand_prob=None, cand_ids=None, diag_offset: int = 0, hard_thr=None, precision: str = "f32"):
    """Forward only (validation loss).  precision "bf16x3": the logits' products on the bf16 MFMA (dim 128 / 256)."""
    _chk_retrieval(q, c, sample_weight, cand_prob, cand_ids, hard_thr, lse, per_row)
    if precision not in SCORER_PRECISIONS:
        raise ValueError(f"precision must be one of {SCORER_PRECISIONS}, got {precision!r}")
    lib = _lib.load()
    fn = lib.tt_retrieval_fwd_f32 if precision == "f32" else lib.tt_retrieval_fwd_bf16x3_f32
    _lib.check(fn(_p(q), _p(c), q.shape[0], c.shape[0], q.shape[1], diag_offset, inv_temperature,
                  _p(sample_weight), _p(cand_prob), _p(cand_ids), _p(hard_thr), _p(workspace),
                  workspace.numel(), _p(lse), _p(per_row), _p(loss), _stream()),
               "tt_retrieval_fwd_f32" if precision == "f32" else "tt_retrieval_fwd_bf16x3_f32")
    return loss


def retrieval_bwd(q, c, inv_temperature: float, workspace, lse, dq, dc, sample_weight=None, cand_prob=None,
                  cand_ids=None, diag_offset: int = 0, grad_scale: float = 1.0, hard_thr=None):
    _chk_retrieval(q, c, sample_weight, cand_prob, cand_ids, hard_thr, lse, None, dq, dc)
    lib = _lib.load()
    _lib.check(lib.tt_retrieval_bwd_f32(_p(q), _p(c), q.shape[0], c.shape[0], q.shape[1], diag_offset, inv_temperature,
                                        _p(sample_weight), _p(cand_prob), _p(cand_ids), _p(hard_thr), _p(lse), grad_scale,
                                        _p(workspace), workspace.numel(), _p(dq), _p(dc), _stream()),
               "tt_retrieval_bwd_f32")
    return dq, dc




def retrieval_fwd_bwd(q, c, inv_temperature: float, workspace, lse, per_row, loss, dq, dc, sample_weight=None,
                      cand_prob=None, cand_ids=None, diag_offset: int = 0, grad_scale: float = 1.0, hard_thr=None,
                      precision: str = "f32"):
    """Loss and both gradients in two fused passes (training form).  precision "f32": exact f32 products on the
    f32-input MFMA; "bf16x3": the f32-emulated split-bf16 form on the bf16 MFMA (dim 128 / 256)."""
    _chk_retrieval(q, c, sample_weight, cand_prob, cand_ids, hard_thr, lse, per_row, dq, dc)
    if precision not in SCORER_PRECISIONS:
        raise ValueError(f"precision must be one of {SCORER_PRECISIONS}, got {precision!r}")
    lib = _lib.load()
    fn = lib.tt_retrieval_fwd_bwd_f32 if precision == "f32" else lib.tt_retrieval_fwd_bwd_bf16x3_f32
    _lib.check(fn(_p(q), _p(c), q.shape[0], c.shape[0], q.shape[1], diag_offset, inv_temperature,
                                            _p(sample_weight), _p(cand_prob), _p(cand_ids), _p(hard_thr), grad_scale, _p(workspace),
              workspace.numel(), _p(lse), _p(per_row), _p(loss), _p(dq), _p(dc), _stream()),
               "tt_retrieval_fwd_bwd_f32" if precision == "f32" else "tt_retrieval_fwd_bwd_bf16x3_f32")
    return loss


def retrieval_rank(q, c, inv_temperature: float, pos_index, workspace=None, cand_prob=None, out=None, precision: str = "f32"):
    """rank[i] = #candidates scoring strictly above query i's true candidate ``pos_index[i]`` (int32 [nq]).
    precision "bf16x3": the logits' products on the bf16 MFMA (dim 128 / 256)."""
    if precision not in SCORER_PRECISIONS:
        raise ValueError(f"precision must be one of {SCORER_PRECISIONS}, got {precision!r}")
    _chk(q, torch.float32, "query_embeddings", 2)
    _chk(c, torch.float32, "candidate_embeddings", 2)
    _chk(pos_index, torch.int64, "pos_index", 1)
    if cand_prob is not None:
        _chk(cand_prob, torch.float32, "candidate_sampling_probability", 1)
    nq, nc, d = q.shape[0], c.shape[0], q.shape[1]
    if q.shape[1] != c.shape[1]:
        raise RuntimeError(f"retrieval: embedding dims differ: {q.shape[1]} vs {c.shape[1]}")
    if pos_index.numel() != nq or (cand_prob is not None and cand_prob.numel() != nc):
        raise RuntimeError("retrieval_rank: pos_index needs nq entries and candidate_sampling_probability nc")
    if workspace is None:       # the rank pass needs only the bias / threshold / count regions, not the gradient slabs
        workspace = torch.empty(retrieval_rank_workspace_bytes(nq, nc, d), dtype=torch.uint8, device=q.device)
    if out is None:
        out = torch.empty(nq, dtype=torch.int32, device=q.device)
    lib = _lib.load()
    fn = lib.tt_retrieval_rank_f32 if precision == "f32" else lib.tt_retrieval_rank_bf16x3_f32
    _lib.check(fn(_p(q), _p(c), nq, nc, d, inv_temperature, _p(cand_prob), _p(pos_index),
                  _p(workspace), workspace.numel(), _p(out), _stream()),
               "tt_retrieval_rank_f32" if precision == "f32" else "tt_retrieval_rank_bf16x3_f32")
    return out


def retrieval_batch_rank(q, c, inv_temperature: float, cand_prob=None, cand_ids=None, diag_offset: int = 0, workspace=None, out=None):
    """In-batch rank of every query's positive (candidate i + diag_offset) under the scores the loss sees - temperature,
    sampling-probability correction, accidental hits removed (int32 [nq]); top-k accuracy = mean(rank < k)."""
    _chk_retrieval(q, c, None, cand_prob, cand_ids)
    nq, nc, d = q.shape[0], c.shape[0], q.shape[1]
    if workspace is None:
        workspace = torch.empty(retrieval_rank_workspace_bytes(nq, nc, d), dtype=torch.uint8, device=q.device)
    if out is None:
        out = torch.empty(nq, dtype=torch.int32, device=q.device)
    lib = _lib.load()
    _lib.check(lib.tt_retrieval_batch_rank_f32(_p(q), _p(c), nq, nc, d, diag_offset, inv_temperature, _p(cand_prob), _p(cand_ids),
                                               _p(workspace), workspace.numel(), _p(out), _stream()), "tt_retrieval_batch_rank_f32")
    return out


def retrieval_hard_negative_thresholds(q, c, inv_temperature: float, k: int, workspace, cand_prob=None, cand_ids=None,
                                       diag_offset: int = 0, scratch=None, out=None):
    """Per-query thresholds for ``num_hard_negatives = k`` (pass as ``hard_thr`` to the loss entry points)."""
    _chk(q, torch.float32, "query_embeddings", 2)
    _chk(c, torch.float32, "candidate_embeddings", 2)
    nq, nc, d = q.shape[0], c.shape[0], q.shape[1]
    if scratch is None:
        scratch = torch.empty(nq * nc, dtype=torch.float32, device=q.device)
    if out is None:
        out = torch.empty(nq, dtype=torch.float32, device=q.device)
    lib = _lib.load()
    _lib.check(lib.tt_retrieval_hard_negative_thresholds_f32(_p(q), _p(c), nq, nc, d, diag_offset, inv_temperature,
                                                             _p(cand_prob), _p(cand_ids), k, _p(workspace), workspace.numel(),
                                                             _p(scratch), scratch.numel() * 4, _p(out), _stream()),
               "tt_retrieval_hard_negative_thresholds_f32")
    return out
