"""CPU restatement of the train step with stock multi-threaded torch CPU ops — the reported
``cpu_baseline`` (kind "port": restatement, NOT reference code).

TEST / BENCH INFRASTRUCTURE (see oracle/__init__.py).  Same semantics as oracle/two_tower.py
(checked against it in tests/test_oracle_cpu.py): F.embedding -> F.linear(+ReLU) ->
q @ c.T / T -> logsumexp - diagonal (SUM) -> autograd -> de-duplicated sparse SGD/Adagrad ->
dense SGD/Adagrad with the l2 term.  There is no reference CPU path to time: the reference's
``src/training`` is a docstring stub (``/root/reference/src/training/__init__.py:1``).
"""
from __future__ import annotations

import time

import torch
import torch.nn.functional as F


class TorchCpuTwoTower:
    def __init__(self, user_table, item_table, user_w, user_b, item_w, item_b, temperature=0.1, l2=1e-6,
                 lr=0.001, optimizer="sgd", acc0=0.1, eps=1e-7):
        self.ut, self.it = user_table, item_table                # [N, D] f32 CPU tensors, updated in place
        self.towers = [[(w.requires_grad_(), b.requires_grad_()) for w, b in zip(user_w, user_b)],
                       [(w.requires_grad_(), b.requires_grad_()) for w, b in zip(item_w, item_b)]]
        self.T, self.l2, self.lr, self.opt, self.eps = temperature, l2, lr, optimizer, eps
        if optimizer == "adagrad":
            self.uacc = torch.full_like(user_table, acc0)
            self.iacc = torch.full_like(item_table, acc0)
            self.dacc = [[(torch.full_like(w, acc0), torch.full_like(b, acc0)) for w, b in tw] for tw in self.towers]

    def _tower(self, x, layers):
        n = len(layers)
        for l, (w, b) in enumerate(layers):
            x = F.linear(x, w.t(), b)                            # w is [in, out] (Keras layout)
            if l < n - 1:
                x = F.relu(x)
        return x

    def _sparse_update(self, table, acc, ids, g):
        uniq, inv = torch.unique(ids, return_inverse=True)
        gs = torch.zeros(uniq.numel(), g.shape[1]).index_add_(0, inv, g)
        if self.opt == "sgd":
            table.index_add_(0, uniq, gs, alpha=-self.lr)
        else:
            a = acc[uniq] + gs * gs
            acc[uniq] = a
            table[uniq] = table[uniq] - (self.lr * gs) / torch.sqrt(a + self.eps)

    def step(self, uid, iid):
        ue = F.embedding(uid, self.ut).requires_grad_()
        ie = F.embedding(iid, self.it).requires_grad_()
        q = self._tower(ue, self.towers[0])
        c = self._tower(ie, self.towers[1])
        s = q @ c.t() / self.T
        loss = (torch.logsumexp(s, dim=1) - s.diagonal()).sum()
        params = [p for tw in self.towers for wb in tw for p in wb]
        grads = torch.autograd.grad(loss, [ue, ie] + params)
        with torch.no_grad():
            self._sparse_update(self.ut, getattr(self, "uacc", None), uid, grads[0])
            self._sparse_update(self.it, getattr(self, "iacc", None), iid, grads[1])
            k = 2
            for t, tw in enumerate(self.towers):
                for l, (w, b) in enumerate(tw):
                    gw = grads[k] + 2 * self.l2 * w
                    gb = grads[k + 1]
                    k += 2
                    if self.opt == "sgd":
                        w -= self.lr * gw
                        b -= self.lr * gb
                    else:
                        aw, ab = self.dacc[t][l]
                        aw += gw * gw
                        ab += gb * gb
                        w -= self.lr * gw / torch.sqrt(aw + self.eps)
                        b -= self.lr * gb / torch.sqrt(ab + self.eps)
        return float(loss.detach())


def time_cpu_steps(model: TorchCpuTwoTower, batches, budget_s: float = 15.0, min_steps: int = 3):
    """Runs steps over ``batches`` (list of (uid, iid)) cyclically until ~budget_s of CPU work;
    returns (seconds_per_step, steps_timed)."""
    model.step(*batches[0])                                      # warm-up (thread pools, allocator)
    t0 = time.perf_counter()
    model.step(*batches[1 % len(batches)])
    one = time.perf_counter() - t0
    n = max(min_steps, min(200, int(budget_s / max(one, 1e-6))))
    t0 = time.perf_counter()
    for i in range(n):
        model.step(*batches[(i + 2) % len(batches)])
    return (time.perf_counter() - t0) / n, n
