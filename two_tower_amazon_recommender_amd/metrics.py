"""Retrieval metrics over a candidate corpus — the role of ``tfrs.metrics.FactorizedTopK`` for the keys the
reference names (``/root/reference/configs/data_config.yaml:71`` ``top_k_eval: [1, 5, 10, 20, 50, 100]``;
``README.md:80`` quotes Recall@10 / NDCG@10).  The rank of every query's true candidate is computed by one fused
pass of the scorer kernel over all candidates (``tt_retrieval_rank_f32``): no [queries x corpus] score matrix and
no top-k extraction — Recall@K = mean(rank < K), NDCG@K = mean([rank < K] / log2(rank + 2)).
"""
from __future__ import annotations

import torch

from . import ops


class FactorizedTopK:
    def __init__(self, ks=(1, 5, 10, 20, 50, 100), temperature: float | None = None, candidates: torch.Tensor | None = None,
                 precision: str = "f32"):
        """candidates: optional [n_candidates, D] corpus embeddings (tfrs.metrics.FactorizedTopK(candidates=...)); then
        ``update_state(q, None, true_index)`` and ``tasks.Retrieval(metrics=...)`` rank against it."""
        self.ks = tuple(int(k) for k in ks)
        self.precision = precision        # "bf16x3": the corpus pass on the bf16 matrix cores (scorer dim 128 / 256)
        self.inv_t = 1.0 if temperature is None else 1.0 / temperature
        self.candidates = None if candidates is None else candidates.contiguous()
        self.reset_state()

    def reset_state(self):
        self._ws = getattr(self, "_ws", None)       # (key, buffer) of the rank pass, kept across batches and resets
        self._n = 0
        self._hits = None
        self._dcg = None

    def update_state(self, query_embeddings: torch.Tensor, candidate_embeddings: torch.Tensor,
                     true_candidate_index: torch.Tensor, candidate_sampling_probability=None) -> torch.Tensor:
        """Accumulates the metrics of one query batch against the candidate corpus; returns the int32 ranks."""
        if candidate_embeddings is None:
            candidate_embeddings = self.candidates
        if candidate_embeddings is None:
            raise ValueError("FactorizedTopK.update_state: no candidate corpus (pass it here or to the constructor)")
        q, c = query_embeddings.contiguous(), candidate_embeddings.contiguous()
        key = (q.shape[0], c.shape[0], q.shape[1], str(q.device))
        if self._ws is None or self._ws[0] != key:         # one small buffer per shape, not one allocation per batch
            self._ws = (key, torch.empty(ops.retrieval_rank_workspace_bytes(*key[:3]), dtype=torch.uint8, device=q.device))
        rank = ops.retrieval_rank(q, c, self.inv_t, true_candidate_index.contiguous(), workspace=self._ws[1],
                                  cand_prob=candidate_sampling_probability, precision=self.precision)
        r = rank.to(torch.float64)
        ks = torch.tensor(self.ks, dtype=torch.float64, device=r.device)
        inside = (r[:, None] < ks[None, :]).to(torch.float64)                   # [nq, len(ks)]
        gain = 1.0 / torch.log2(r + 2.0)
        hits, dcg = inside.sum(0), (inside * gain[:, None]).sum(0)
        self._hits = hits if self._hits is None else self._hits + hits
        self._dcg = dcg if self._dcg is None else self._dcg + dcg
        self._n += r.numel()
        return rank

    def result(self) -> dict:
        if not self._n:
            return {}
        hits, dcg = (self._hits / self._n).tolist(), (self._dcg / self._n).tolist()
        out = {}
        for k, h, d in zip(self.ks, hits, dcg):
            out[f"recall@{k}"] = h
            out[f"ndcg@{k}"] = d          # one relevant item per query: the ideal DCG is 1
        return out


class TopKCategoricalAccuracy:
    """In-batch top-k accuracy - the role of ``tf.keras.metrics.TopKCategoricalAccuracy(k)`` in
    ``tfrs.tasks.Retrieval(batch_metrics=[...])``: the share of queries whose positive is among the k highest-scoring
    candidates OF THE BATCH (ties at the boundary count as inside, like ``tf.math.in_top_k``).  It consumes the ranks the
    fused metric pass produces (``torch.ops.twotower.retrieval_batch_rank``): no [queries x candidates] score matrix."""

    def __init__(self, k: int = 5, name: str | None = None):
        if k < 1:
            raise ValueError("k must be >= 1")
        self.k = int(k)
        self.name = name or f"top_{self.k}_categorical_accuracy"
        self.reset_state()

    def reset_state(self):
        self._hits = None
        self._n = 0

    def update_state_from_ranks(self, rank: torch.Tensor, sample_weight: torch.Tensor | None = None):
        inside = (rank < self.k).to(torch.float64)
        if sample_weight is not None:                    # Keras: a weighted mean
            w = sample_weight.to(torch.float64)
            hits, n = (inside * w).sum(), w.sum()
        else:
            hits, n = inside.sum(), torch.tensor(float(rank.numel()), dtype=torch.float64, device=rank.device)
        self._hits = hits if self._hits is None else self._hits + hits
        self._n = n if isinstance(self._n, int) and self._n == 0 else self._n + n

    def result(self) -> float:
        if self._hits is None:
            return 0.0
        return float((self._hits / self._n).item())

