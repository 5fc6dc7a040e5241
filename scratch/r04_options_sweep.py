"""Round 4: step time with the Retrieval task's options on (cfg3 shapes, batch 8192): sample weights, sampling-probability correction,
accidental-hit removal (candidate ids), hard negatives."""
import json, sys, time
import torch
sys.path.insert(0, ".")
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer
dev = torch.device("cuda:0")
b = 8192
def run(name, cfg_kw, step_kw_fn):
    cfg = TwoTowerConfig(n_users=2_000_000, n_items=1_000_000, embedding_dim=128, tower_dims=[256, 128], temperature=0.1,
                         l2_regularization=1e-6, learning_rate=0.001, optimizer="sgd", batch_size=b, **cfg_kw)
    tr = TwoTowerTrainer(cfg, dev, seed=1)
    batches = [tr.synthetic_batch(1, s) for s in range(8)]
    kws = [step_kw_fn(tr, batches[s]) for s in range(8)]
    for s in range(30): tr.step(*batches[s % 8], **kws[s % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 150
    for s in range(n): tr.step(*batches[s % 8], **kws[s % 8])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"case": name, "ms_per_step": round(ms, 4), "composite": tr._cstep is not None}), flush=True)
    del tr; torch.cuda.empty_cache()
run("plain", {}, lambda tr, bt: {})
run("sample_weight", {}, lambda tr, bt: {"sample_weight": torch.rand(b, device=dev) + 0.5})
run("candidate_sampling_probability", {}, lambda tr, bt: {"candidate_sampling_probability": torch.rand(b, device=dev) * 0.01 + 1e-4})
run("candidate_ids (accidental hits)", {}, lambda tr, bt: {"candidate_ids": bt[1]})
run("all three", {}, lambda tr, bt: {"sample_weight": torch.rand(b, device=dev) + 0.5, "candidate_sampling_probability": torch.rand(b, device=dev) * 0.01 + 1e-4, "candidate_ids": bt[1]})
try:
    run("num_hard_negatives 100", {"num_hard_negatives": 100}, lambda tr, bt: {})
    run("num_hard_negatives 100 + candidate_ids", {"num_hard_negatives": 100}, lambda tr, bt: {"candidate_ids": bt[1]})
except Exception as e:
    print(json.dumps({"case": "hard negatives", "error": str(e)[:200]}))
