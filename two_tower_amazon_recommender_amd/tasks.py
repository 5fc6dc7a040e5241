"""``Retrieval`` — the task object of the hot path, with the call signature of
``tfrs.tasks.Retrieval`` (tensorflow-recommenders 0.7.x; the reference declares the dependency at
``/root/reference/pyproject.toml:24`` and the settings at ``configs/data_config.yaml:68-71`` but never
calls it).  The scorer, the in-batch sampled-softmax loss and its gradient run in ONE family of fused
HIP kernels (csrc/score.hip); softmax probabilities are never stored (the exact-f32 training entry keeps the raw
[num_queries, num_candidates] dot products in its workspace between its two passes; the forward-only form keeps nothing).

The task dispatches through the registered custom ops ``torch.ops.twotower.retrieval_loss`` (training: loss and
both gradients in the fused two-pass form, 6*Bq*Bc*D executed FLOPs, gradients saved for autograd) and
``torch.ops.twotower.retrieval_loss_value`` (no gradient needed: one statistics pass) — ``torch_ops.py``.

Differences from TFRS, all loud:
  * ``loss`` must be None or ``tasks.CategoricalCrossentropy(from_logits=True, reduction="sum")`` - the TFRS default, the
    one loss the fused kernels implement; anything else (hinge / pairwise / MSE losses of tfrs.losses, or another
    reduction) needs the [queries x candidates] score matrix and raises NotImplementedError;
  * ``metrics`` takes a ``metrics.FactorizedTopK`` built WITH its candidate corpus (``FactorizedTopK(candidates=...)``);
    it is updated when ``compute_metrics`` is true, and ``candidate_ids`` must then be the int64 index of every
    candidate in that corpus (the reference's ``item_idx``); ``loss_metrics`` takes objects with
    ``update_state(loss)``; ``batch_metrics`` takes ``metrics.TopKCategoricalAccuracy(k)`` objects (the Keras metric TFRS
    users pass there): in-batch top-k accuracy is ``rank < k`` from one fused rank pass over the batch's candidates, under
    the scores the loss sees (temperature, sampling-probability correction, accidental hits removed) - no score matrix;
    any other Keras metric would need the [queries x candidates] matrix and raises TypeError.  As in TFRS
    (``metric.update_state(labels, scores)``) they are NOT weighted by ``sample_weight``.  TFRS computes them on the scores
    AFTER hard-negative mining; the rank pass sees every in-batch candidate, so ``batch_metrics`` together with
    ``num_hard_negatives`` raises NotImplementedError instead of reporting a different number;
  * ``num_hard_negatives=k`` keeps the positive and the k highest-scoring negatives per query
    (tfrs.layers.loss.HardNegativeMining); negatives tied with the k-th are all kept;
  * ``candidate_ids`` must be an int64 tensor (the reference's ids are int64:
    ``prepare_training_data.py:209-210``).
"""
from __future__ import annotations

import torch

from . import torch_ops  # noqa: F401  (registers torch.ops.twotower.*)


class CategoricalCrossentropy:
    """Marker for the TFRS default loss (``tf.keras.losses.CategoricalCrossentropy(from_logits=True, reduction=SUM)``)."""

    def __init__(self, from_logits: bool = True, reduction: str = "sum"):
        self.from_logits, self.reduction = bool(from_logits), str(reduction).lower()


class Retrieval:
    """A factorized retrieval task: in-batch softmax over query x candidate dot products."""

    def __init__(self, loss=None, metrics=None, batch_metrics=None, loss_metrics=None, temperature=None,
                 num_hard_negatives=None, remove_accidental_hits=False, name="retrieval_task", precision="f32"):
        if loss is not None and not (isinstance(loss, CategoricalCrossentropy) and loss.from_logits and loss.reduction == "sum"):
            raise NotImplementedError("Retrieval(loss=...): only the TFRS default loss - tasks.CategoricalCrossentropy("
                                      "from_logits=True, reduction='sum') - is implemented in the HIP path; hinge, pairwise "
                                      "and pointwise losses (and other reductions) need the [queries x candidates] score "
                                      "matrix, which the fused kernels never materialise")
        from .metrics import TopKCategoricalAccuracy
        self._batch_metrics = list(batch_metrics) if batch_metrics is not None else []
        for m in self._batch_metrics:
            if not isinstance(m, TopKCategoricalAccuracy):
                raise TypeError("Retrieval(batch_metrics=...) takes metrics.TopKCategoricalAccuracy(k) objects (in-batch top-k "
                                "accuracy from the fused rank pass); other metrics would need the in-batch score matrix")
        if metrics is not None:
            from .metrics import FactorizedTopK
            if not isinstance(metrics, FactorizedTopK) or metrics.candidates is None:
                raise TypeError("Retrieval(metrics=...) takes a metrics.FactorizedTopK built with its candidate corpus "
                                "(FactorizedTopK(candidates=item_corpus_embeddings))")
        if self._batch_metrics and num_hard_negatives is not None:
            raise NotImplementedError("Retrieval(batch_metrics=..., num_hard_negatives=k): TFRS evaluates batch metrics on the scores "
                                      "after hard-negative mining; the fused rank pass ranks against every in-batch candidate")
        self._factorized_metrics = metrics
        self._loss_metrics = list(loss_metrics) if loss_metrics is not None else []
        for m in self._loss_metrics:
            if not hasattr(m, "update_state"):
                raise TypeError("Retrieval(loss_metrics=...): every entry needs update_state(loss)")
        if num_hard_negatives is not None and num_hard_negatives < 1:
            raise ValueError("num_hard_negatives must be a positive integer")
        if temperature is not None and temperature <= 0:
            raise ValueError("temperature must be positive")
        self._temperature = temperature
        self._num_hard_negatives = num_hard_negatives
        self._remove_accidental_hits = remove_accidental_hits
        if precision not in ("f32", "bf16x3"):
            raise ValueError("precision must be 'f32' or 'bf16x3'")
        self._precision = precision          # (extension) matrix-product precision of the training passes: torch_ops.py
        self.name = name
        self.last_per_example_loss = None

    @property
    def factorized_metrics(self):
        return self._factorized_metrics

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
        k = 0 if self._num_hard_negatives is None else int(self._num_hard_negatives)
        if torch.is_grad_enabled() and (q.requires_grad or c.requires_grad):
            loss, per_example, _, _ = torch.ops.twotower.retrieval_loss(q, c, sw, cp, ids, inv_t, diag_offset, k, self._precision)
        else:
            loss, per_example = torch.ops.twotower.retrieval_loss_value(q, c, sw, cp, ids, inv_t, diag_offset, k, self._precision)
        self.last_per_example_loss = per_example
        if compute_metrics and self._factorized_metrics is not None:
            if candidate_ids is None:
                raise ValueError("Retrieval(metrics=FactorizedTopK): pass candidate_ids = the index of every candidate "
                                 "in the metric's corpus")
            with torch.no_grad():
                true_index = candidate_ids[diag_offset:diag_offset + q.shape[0]]
                self._factorized_metrics.update_state(q.detach(), None, true_index)
        if compute_batch_metrics and self._batch_metrics:
            with torch.no_grad():
                rank = torch.ops.twotower.retrieval_batch_rank(q.detach(), c.detach(), cp, ids, inv_t, diag_offset)
                for m in self._batch_metrics:
                    m.update_state_from_ranks(rank)              # unweighted, as TFRS: only the loss and loss_metrics see sample_weight
        for m in self._loss_metrics:
            m.update_state(loss.detach())
        return loss

    @property
    def batch_metrics(self):
        return self._batch_metrics

    call = __call__
