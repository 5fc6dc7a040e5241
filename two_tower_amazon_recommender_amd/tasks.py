"""``Retrieval`` — the task object of the hot path, with the call signature of
``tfrs.tasks.Retrieval`` (tensorflow-recommenders 0.7.x; the reference declares the dependency at
``/root/reference/pyproject.toml:24`` and the settings at ``configs/data_config.yaml:68-71`` but never
calls it).  The scorer, the in-batch sampled-softmax loss and its gradient run in ONE family of fused
HIP kernels (csrc/score.hip): the [num_queries, num_candidates] logits never reach HBM.

Differences from TFRS, all loud:
  * ``loss`` must be None (the TFRS default: CategoricalCrossentropy(from_logits=True, reduction=SUM));
  * ``metrics`` / ``batch_metrics`` / ``loss_metrics`` objects are not accepted by the task (use
    ``metrics.FactorizedTopK`` directly: it runs the fused rank pass) and raise NotImplementedError;
  * ``num_hard_negatives=k`` keeps the positive and the k highest-scoring negatives per query
    (tfrs.layers.loss.HardNegativeMining); negatives tied with the k-th are all kept;
  * ``candidate_ids`` must be an int64 tensor (the reference's ids are int64:
    ``prepare_training_data.py:209-210``).
"""
from __future__ import annotations

import torch

from . import ops


class _RetrievalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, c, sample_weight, cand_prob, cand_ids, inv_t, diag_offset, task):
        k = task._num_hard_negatives
        q = q.contiguous()
        c = c.contiguous()
        nq, nc, d = q.shape[0], c.shape[0], q.shape[1]
        ws = task._workspace(nq, nc, d, q.device)
        lse = torch.empty(nq, dtype=torch.float32, device=q.device)
        per_row = torch.empty(nq, dtype=torch.float32, device=q.device)
        loss = torch.empty(1, dtype=torch.float32, device=q.device)
        thr = None
        if k is not None:       # non-differentiable selection of the k hardest negatives per query (as in TFRS)
            thr = ops.retrieval_hard_negative_thresholds(q, c, inv_t, k, ws, cand_prob=cand_prob, cand_ids=cand_ids,
                                                         diag_offset=diag_offset)
        ops.retrieval_fwd(q, c, inv_t, ws, lse, per_row, loss, sample_weight=sample_weight, cand_prob=cand_prob,
                          cand_ids=cand_ids, diag_offset=diag_offset, hard_thr=thr)
        ctx.save_for_backward(q, c, lse, sample_weight, cand_prob, cand_ids, thr)
        ctx.inv_t, ctx.diag_offset, ctx.task = inv_t, diag_offset, task
        task.last_per_example_loss = per_row
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        q, c, lse, sample_weight, cand_prob, cand_ids, thr = ctx.saved_tensors
        ws = ctx.task._workspace(q.shape[0], c.shape[0], q.shape[1], q.device)
        dq = torch.empty_like(q)
        dc = torch.empty_like(c)
        ops.retrieval_bwd(q, c, ctx.inv_t, ws, lse, dq, dc, sample_weight=sample_weight, cand_prob=cand_prob,
                          cand_ids=cand_ids, diag_offset=ctx.diag_offset, hard_thr=thr)
        # upstream gradient of the scalar loss stays on the device (no host sync)
        return dq * grad_out, dc * grad_out, None, None, None, None, None, None


class Retrieval:
    """A factorized retrieval task: in-batch softmax over query x candidate dot products."""

    def __init__(self, loss=None, metrics=None, batch_metrics=None, loss_metrics=None, temperature=None,
                 num_hard_negatives=None, remove_accidental_hits=False, name="retrieval_task"):
        if loss is not None:
            raise NotImplementedError("Retrieval(loss=...): only the TFRS default loss (categorical cross-entropy "
                                      "from logits, SUM reduction) is implemented in the HIP path")
        if metrics is not None or batch_metrics is not None or loss_metrics is not None:
            raise NotImplementedError("Retrieval metrics (FactorizedTopK etc.) are not implemented yet (SURVEY.md §8f)")
        if num_hard_negatives is not None and num_hard_negatives < 1:
            raise ValueError("num_hard_negatives must be a positive integer")
        if temperature is not None and temperature <= 0:
            raise ValueError("temperature must be positive")
        self._temperature = temperature
        self._num_hard_negatives = num_hard_negatives
        self._remove_accidental_hits = remove_accidental_hits
        self.name = name
        self._ws = {}
        self.last_per_example_loss = None

    def _workspace(self, nq, nc, d, device):
        key = (nq, nc, d, str(device))
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()                      # one live shape at a time: the slabs are tens of MB
            ws = torch.empty(ops.retrieval_workspace_bytes(nq, nc, d), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

    def __call__(self, query_embeddings, candidate_embeddings, sample_weight=None, candidate_sampling_probability=None,
                 candidate_ids=None, compute_metrics=True, compute_batch_metrics=True, diag_offset: int = 0):
        """Returns the scalar loss (a 0-d CUDA tensor with autograd).  ``diag_offset`` (extension): the
        positive of query i is candidate i + diag_offset — used by the sharded multi-GPU slab."""
        if self._remove_accidental_hits and candidate_ids is None:
            raise ValueError("When accidental hit removal is enabled, candidate ids must be supplied.")
        q, c = query_embeddings, candidate_embeddings
        if q.dim() != 2 or c.dim() != 2 or q.shape[1] != c.shape[1]:
            raise RuntimeError(f"Retrieval: expected [Bq,D] and [Bc,D] embeddings, got {tuple(q.shape)} and {tuple(c.shape)}")
        inv_t = 1.0 if self._temperature is None else 1.0 / self._temperature
        ids = candidate_ids if self._remove_accidental_hits else None
        sw = None if sample_weight is None else sample_weight.to(torch.float32).contiguous()
        cp = None if candidate_sampling_probability is None else candidate_sampling_probability.to(torch.float32).contiguous()
        return _RetrievalLoss.apply(q, c, sw, cp, ids, inv_t, diag_offset, self)

    call = __call__
