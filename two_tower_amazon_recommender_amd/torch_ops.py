"""``torch.ops.twotower.*`` — the hot-path kernels registered as PyTorch custom ops (``torch.library``), as
``north_star`` asks ("exposed to Python through PyTorch-ROCm custom ops that keep the repo's
tfrs.tasks.Retrieval-style call signature") and SURVEY.md §8(b) row 3 specifies.  Each op is a thin, typed front
over the same C ABI (``include/twotower_hip.h``, reached through ``ops``): schema + fake (meta) implementation +
autograd formula, so the ops compose with ``torch.autograd``, ``torch.compile`` tracing and ``opcheck``.  There
is no CPU implementation: the ops are registered for ``device_types="cuda"`` only and a CPU tensor raises.

    loss, per_example, dq, dc = torch.ops.twotower.retrieval_loss(q, c, w, p, ids, inv_t, diag_offset, k)
    rows = torch.ops.twotower.embedding_gather(table, ids)
    y    = torch.ops.twotower.dense_fwd(x, w, b, relu)                  # autograd through twotower::dense_bwd
    torch.ops.twotower.sparse_update_(table, accum, grads, ids, "adagrad", lr, eps)

What the reference would have run through TensorFlow / TFRS for these (``/root/reference/pyproject.toml:22,24``;
settings ``configs/data_config.yaml:54-71``; the task object's call signature: SURVEY.md Appendix A).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops

NS = "twotower"


_WS_CACHE: dict = {}
_PLAN_CACHE: dict = {}


def _ws(nq: int, nc: int, d: int, device, forward_only: bool = False) -> Tensor:
    """Scorer workspace; the training entry's includes the [nq, nc] f32 dot-product buffer its second pass reads back
    (4*nq*nc bytes: 268 MB at 8192 x 8192, 4.3 GB at 32768 x 32768).  Kept per (kind, device, stream) instead of being
    allocated on every call: the kernels of one stream run in order, so successive calls can share it.  The LARGEST buffer seen
    is kept (a loop alternating two batch shapes re-allocated 268 MB per call when only the last shape was held: VERDICT r03);
    ``release_workspaces()`` drops them.  Nothing is cached while the stream is being captured into a HIP graph: a buffer
    allocated there lives in the graph's private pool and must not be handed to eager calls afterwards."""
    n = ops.retrieval_fwd_workspace_bytes(nq, nc, d) if forward_only else ops.retrieval_workspace_bytes(nq, nc, d)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(n, dtype=torch.uint8, device=device)
    key = (bool(forward_only), str(device), torch.cuda.current_stream(device).cuda_stream)
    hit = _WS_CACHE.get(key)
    if hit is not None and hit.numel() >= n:
        return hit
    _WS_CACHE.pop(key, None)                 # (freed before the larger one is allocated)
    buf = torch.empty(n, dtype=torch.uint8, device=device)
    _WS_CACHE[key] = buf
    return buf


def _plan(n: int, device) -> "ops.SparsePlan":
    """One SparsePlan (sorted ids, positions, piece workspace) per (length, device, stream) for ``sparse_update_``."""
    if torch.cuda.is_current_stream_capturing():
        return ops.SparsePlan(n, device)
    key = (int(n), str(device), torch.cuda.current_stream(device).cuda_stream)
    hit = _PLAN_CACHE.get(key)
    if hit is None:
        hit = _PLAN_CACHE[key] = ops.SparsePlan(n, device)
    return hit


def release_workspaces() -> None:
    """Drop every cached scorer workspace and sort plan of the custom ops (they are re-allocated on the next call)."""
    _WS_CACHE.clear()
    _PLAN_CACHE.clear()


# --------------------------------------------------------------------------------------------- a1 lookup
@torch.library.custom_op(f"{NS}::embedding_gather", mutates_args=(), device_types="cuda")
def embedding_gather(table: Tensor, ids: Tensor) -> Tensor:
    """rows[b, :] = table[ids[b], :] (ids outside the table give a zero row, like the C entry point; use the trainer's
    flag for TF's raise).  Not differentiable: the table is trained by ``sparse_update_`` on the row gradients."""
    return ops.embedding_gather(table.contiguous(), ids.contiguous())


@embedding_gather.register_fake
def _(table, ids):
    return table.new_empty((ids.shape[0], table.shape[1]))


# --------------------------------------------------------------------------------------------- a3 + a4
@torch.library.custom_op(f"{NS}::retrieval_loss", mutates_args=(), device_types="cuda")
def retrieval_loss(query_embeddings: Tensor, candidate_embeddings: Tensor, sample_weight: Optional[Tensor],
                   candidate_sampling_probability: Optional[Tensor], candidate_ids: Optional[Tensor],
                   inv_temperature: float, diag_offset: int, num_hard_negatives: int,
                   precision: str = "f32") -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """In-batch sampled-softmax loss (SUM) of tfrs.tasks.Retrieval AND its gradients, in the fused two-pass form
    (exact f32: 6*Bq*Bc*D executed FLOPs - pass 2 reads the raw dot products of pass 1 back from the workspace;
    bf16x3: both passes compute them).  Returns (loss [], per-example loss [Bq], dLoss/dq, dLoss/dc);
    the autograd formula multiplies the saved gradients by the incoming scalar gradient.  ``candidate_ids`` given =
    accidental-hit removal; ``num_hard_negatives`` > 0 keeps the positive and the k hardest negatives per query;
    ``precision`` "f32" (exact f32 products) or "bf16x3" (f32-emulated split-bf16 products, dim 128 / 256)."""
    q, c = query_embeddings.contiguous(), candidate_embeddings.contiguous()
    nq, nc, d = q.shape[0], c.shape[0], q.shape[1]
    ws = _ws(nq, nc, d, q.device)
    lse, per_row = q.new_empty(nq), q.new_empty(nq)
    loss = q.new_empty(1)
    dq, dc = torch.empty_like(q), torch.empty_like(c)
    thr = None
    if num_hard_negatives > 0:
        thr = ops.retrieval_hard_negative_thresholds(q, c, inv_temperature, num_hard_negatives, ws,
                                                     cand_prob=candidate_sampling_probability, cand_ids=candidate_ids,
                                                     diag_offset=diag_offset)
    ops.retrieval_fwd_bwd(q, c, inv_temperature, ws, lse, per_row, loss, dq, dc, sample_weight=sample_weight,
                          cand_prob=candidate_sampling_probability, cand_ids=candidate_ids, diag_offset=diag_offset,
                          hard_thr=thr, precision=precision)
    return loss.reshape(()), per_row, dq, dc


@retrieval_loss.register_fake
def _(q, c, sample_weight, candidate_sampling_probability, candidate_ids, inv_temperature, diag_offset, num_hard_negatives,
      precision="f32"):
    return q.new_empty(()), q.new_empty((q.shape[0],)), torch.empty_like(q), torch.empty_like(c)


def _retrieval_setup(ctx, inputs, output):
    _, per_example, dq, dc = output
    # only the scalar loss carries a gradient: the per-example losses and the two gradient tensors are results, not
    # differentiable functions here (a loss built from per_example would otherwise train on silent zeros)
    ctx.mark_non_differentiable(per_example, dq, dc)
    ctx.save_for_backward(dq, dc)


def _retrieval_backward(ctx, g_loss, g_per_example, g_dq, g_dc):
    dq, dc = ctx.saved_tensors
    # the upstream gradient of the scalar loss stays on the device (no host sync)
    sdq, sdc = torch._foreach_mul((dq, dc), g_loss)            # (one launch for both)
    return sdq, sdc, None, None, None, None, None, None, None


retrieval_loss.register_autograd(_retrieval_backward, setup_context=_retrieval_setup)


@torch.library.custom_op(f"{NS}::retrieval_loss_value", mutates_args=(), device_types="cuda")
def retrieval_loss_value(query_embeddings: Tensor, candidate_embeddings: Tensor, sample_weight: Optional[Tensor],
                         candidate_sampling_probability: Optional[Tensor], candidate_ids: Optional[Tensor],
                         inv_temperature: float, diag_offset: int, num_hard_negatives: int,
                         precision: str = "f32") -> Tuple[Tensor, Tensor]:
    """Forward only (validation): (loss [], per-example loss [Bq]) in one statistics pass; ``precision`` as in
    ``retrieval_loss`` (a task that trains in bf16x3 validates in bf16x3)."""
    q, c = query_embeddings.contiguous(), candidate_embeddings.contiguous()
    nq, nc, d = q.shape[0], c.shape[0], q.shape[1]
    ws = _ws(nq, nc, d, q.device, forward_only=True)
    lse, per_row = q.new_empty(nq), q.new_empty(nq)
    loss = q.new_empty(1)
    thr = None
    if num_hard_negatives > 0:
        thr = ops.retrieval_hard_negative_thresholds(q, c, inv_temperature, num_hard_negatives, ws,
                                                     cand_prob=candidate_sampling_probability, cand_ids=candidate_ids,
                                                     diag_offset=diag_offset)
    ops.retrieval_fwd(q, c, inv_temperature, ws, lse, per_row, loss, sample_weight=sample_weight,
                      cand_prob=candidate_sampling_probability, cand_ids=candidate_ids, diag_offset=diag_offset, hard_thr=thr,
                      precision=precision)
    return loss.reshape(()), per_row


@retrieval_loss_value.register_fake
def _(q, c, sample_weight, candidate_sampling_probability, candidate_ids, inv_temperature, diag_offset, num_hard_negatives,
      precision="f32"):
    return q.new_empty(()), q.new_empty((q.shape[0],))


@torch.library.custom_op(f"{NS}::retrieval_rank", mutates_args=(), device_types="cuda")
def retrieval_rank(query_embeddings: Tensor, candidate_embeddings: Tensor, true_candidate_index: Tensor,
                   candidate_sampling_probability: Optional[Tensor], inv_temperature: float) -> Tensor:
    """rank[i] = number of candidates scoring strictly above query i's true candidate (int32): Recall@K / NDCG@K."""
    return ops.retrieval_rank(query_embeddings.contiguous(), candidate_embeddings.contiguous(), inv_temperature,
                              true_candidate_index.contiguous(), cand_prob=candidate_sampling_probability)


@retrieval_rank.register_fake
def _(q, c, true_candidate_index, candidate_sampling_probability, inv_temperature):
    return q.new_empty((q.shape[0],), dtype=torch.int32)


@torch.library.custom_op(f"{NS}::retrieval_batch_rank", mutates_args=(), device_types="cuda")
def retrieval_batch_rank(query_embeddings: Tensor, candidate_embeddings: Tensor, candidate_sampling_probability: Optional[Tensor],
                         candidate_ids: Optional[Tensor], inv_temperature: float, diag_offset: int) -> Tensor:
    """In-batch rank of every query's positive under the scores the loss sees (int32): batch top-k accuracy = mean(rank < k)."""
    return ops.retrieval_batch_rank(query_embeddings.contiguous(), candidate_embeddings.contiguous(), inv_temperature,
                                    cand_prob=candidate_sampling_probability, cand_ids=candidate_ids, diag_offset=diag_offset)


@retrieval_batch_rank.register_fake
def _(q, c, candidate_sampling_probability, candidate_ids, inv_temperature, diag_offset):
    return q.new_empty((q.shape[0],), dtype=torch.int32)


# --------------------------------------------------------------------------------------------- a2 dense layers
@torch.library.custom_op(f"{NS}::dense_fwd", mutates_args=(), device_types="cuda")
def dense_fwd(x: Tensor, w: Tensor, b: Optional[Tensor], relu: bool) -> Tensor:
    """y = act(x @ w + b)  (Keras Dense, [in, out] kernel; f32-input MFMA)."""
    return ops.dense_fwd(x.contiguous(), w.contiguous(), None if b is None else b.contiguous(), relu)


@dense_fwd.register_fake
def _(x, w, b, relu):
    return x.new_empty((x.shape[0], w.shape[1]))


@torch.library.custom_op(f"{NS}::dense_bwd", mutates_args=(), device_types="cuda")
def dense_bwd(x: Tensor, w: Tensor, dy: Tensor, y: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """(dx, dw, db) of y = act(x @ w + b) given dL/dy; ``y`` (the layer's ReLU output) masks dy, None = linear layer.
    dw / db: split-K slabs over the batch summed in slab order (bitwise reproducible)."""
    x, w, dy = x.contiguous(), w.contiguous(), dy.contiguous()
    m, k, n = x.shape[0], x.shape[1], w.shape[1]
    dz = dy if y is None else torch.ops.aten.threshold_backward(dy, y, 0.0)    # dy where y > 0 else 0: ONE kernel (gt + where: two)
    ns = ops.dense_bwd_num_slabs(m)
    dx = torch.empty_like(x)
    dw_slabs, db_slabs = x.new_empty(ns, k, n), x.new_empty(ns, n)
    ops.dense_bwd(x, w, dz, dx, None, dw_slabs, db_slabs)
    dw, db = x.new_empty(k, n), x.new_empty(n)
    segs = [ops.make_dense_seg(dw, None, dw_slabs, ns, 0.0, dw), ops.make_dense_seg(db, None, db_slabs, ns, 0.0, db)]
    ops.dense_update_(segs, "sgd", 0.0, apply=False)          # slab sums only (grad_out), nothing is updated
    return dx, dw, db


@dense_bwd.register_fake
def _(x, w, dy, y):
    return torch.empty_like(x), torch.empty_like(w), x.new_empty((w.shape[1],))


def _dense_setup(ctx, inputs, output):
    x, w, b, relu = inputs
    ctx.save_for_backward(x, w, output if relu else None)
    ctx.has_bias = b is not None


def _dense_backward(ctx, dy):
    x, w, y = ctx.saved_tensors
    dx, dw, db = torch.ops.twotower.dense_bwd(x, w, dy, y)
    return dx, dw, (db if ctx.has_bias else None), None


dense_fwd.register_autograd(_dense_backward, setup_context=_dense_setup)


# --------------------------------------------------------------------------------------------- a5 sparse optimizer
@torch.library.custom_op(f"{NS}::sparse_update_", mutates_args=("table", "accum"), device_types="cuda")
def sparse_update_(table: Tensor, accum: Optional[Tensor], grads: Tensor, ids: Tensor, optimizer: str, lr: float,
                   eps: float) -> None:
    """Fused sparse SGD / Keras-2.15 Adagrad on the rows ``ids`` of ``table`` (in place; duplicates summed first, in
    ascending position order): one sort launch + one apply launch."""
    if optimizer not in ("sgd", "adagrad"):
        raise ValueError(f"optimizer must be 'sgd' or 'adagrad', got {optimizer!r}")
    plan = _plan(ids.numel(), ids.device).run(ids.contiguous(), table.shape[0])
    if optimizer == "sgd":
        ops.sparse_sgd_(table, grads.contiguous(), plan, lr)
    else:
        if accum is None:
            raise ValueError("sparse_update_: Adagrad needs the accumulator tensor")
        ops.sparse_adagrad_(table, accum, grads.contiguous(), plan, lr, eps)


OPS = ("embedding_gather", "retrieval_loss", "retrieval_loss_value", "retrieval_rank", "retrieval_batch_rank", "dense_fwd", "dense_bwd",
       "sparse_update_")
