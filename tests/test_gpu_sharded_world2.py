"""Two ranks of the row-sharded trainer on ONE GPU (both processes use cuda:0): every HIP kernel of the N>1 path runs
with world = 2 — routing, owner gather, row/gradient exchange buffers, owner-side sort + fused update with duplicates
across ranks, dense all-reduce — and the result is compared with the f64 oracle on the GLOBAL batch.  RCCL refuses
two ranks on one device, so the collectives go through gloo on host copies (a test-local shim around
torch.distributed); the exchange CODE under test is the product's."""
import os
import pathlib
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Done:
    def wait(self):
        return True


def _install_host_bounce():
    real_a2a, real_ar = dist.all_to_all_single, dist.all_reduce

    def a2a(out, inp, group=None, async_op=False, **kw):
        torch.cuda.synchronize()
        o = torch.empty(out.shape, dtype=out.dtype)
        real_a2a(o, inp.cpu(), group=group)
        out.copy_(o)
        return _Done()

    def ar(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        torch.cuda.synchronize()
        c = t.cpu()
        real_ar(c, op=op, group=group)
        t.copy_(c)
        return _Done()

    real_ag, real_rs = dist.all_gather_into_tensor, dist.reduce_scatter_tensor

    def ag(out, inp, group=None, async_op=False):
        torch.cuda.synchronize()
        o = torch.empty(out.shape, dtype=out.dtype)
        real_ag(o, inp.cpu(), group=group)
        out.copy_(o)
        return _Done()

    def rs(out, inp, op=dist.ReduceOp.SUM, group=None, async_op=False):
        torch.cuda.synchronize()
        full = inp.cpu()
        real_ar(full, op=op, group=group)                         # gloo has no reduce_scatter: all-reduce, keep my slice
        n = out.shape[0]
        out.copy_(full[dist.get_rank(group) * n:(dist.get_rank(group) + 1) * n])
        return _Done()

    dist.all_to_all_single, dist.all_reduce = a2a, ar
    dist.all_gather_into_tensor, dist.reduce_scatter_tensor = ag, rs


def _init(rank, world, port, real):
    """real=False: two ranks share cuda:0, collectives bounced through gloo.  real=True: one GPU per rank, the product's
    own RCCL collectives with their real stream ordering (needs >= world GPUs); objects travel over a gloo side group."""
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if real:
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        return dev, dist.new_group(backend="gloo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_host_bounce()
    return torch.device("cuda:0"), None


def _worker(rank, world, port, opt, variant, negatives, nb, ret, real=False):
    try:
        dev, og = _init(rank, world, port, real)
        from oracle import synth, two_tower as tt
        from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer
        from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig
        n_users, n_items, dim, tower_dims, b, seed = 3001, 2000, 64, [128, 64], 1024, 41
        cfg = TwoTowerConfig(n_users=n_users, n_items=n_items, embedding_dim=dim, tower_dims=tower_dims, temperature=0.1,
                             l2_regularization=1e-6, learning_rate=0.001, optimizer=opt, batch_size=b, n_category_buckets=nb)
        tr = ShardedTwoTowerTrainer(cfg, dev, seed=seed, negatives=negatives, capacity_factor=3.0)
        ref = tt.synthetic_state(seed, n_users, n_items, dim, tower_dims, dtype=np.float64, optimizer=opt,
                                 n_category_buckets=nb)

        def cats(step, r):          # category buckets of rank r's sub-batch (None when the model has no such feature)
            return synth.batch_ids(seed, synth.TID_CATEGORY_IDS, step * world + r, b, nb, "Z") if nb else None
        # each rank holds exactly its rows of the synthetic tables
        assert np.array_equal(tr.user_table.cpu().numpy(),
                              ref.user_table[rank::world].astype(np.float32))
        batches = [tr.synthetic_batch(seed, step, variant) for step in range(2)]
        for step in range(2):
            u, i = batches[step]
            # step 0 hands step 1's ids over: their routing and id exchange are issued in the middle of step 0
            dc = tr.synthetic_categories(seed, step) if nb else None
            if nb:
                assert np.array_equal(dc.cpu().numpy(), cats(step, rank))
            loss = tr.step(u, i, next_ids=batches[1] if step == 0 else None, category_ids=dc).item()
            assert (tr.emb._cur == 1) == (step == 1)
            tr.check_ids()
            # oracle: forward/backward of EVERY rank's sub-batch on the shared state, then one update with all gradients
            subs, dense = [], None
            if negatives == "global":
                # every query sees all world*b candidates: identical to ONE oracle step on the concatenated batch
                ids_u = np.concatenate([synth.batch_ids(seed, synth.TID_USER_IDS, step * world + r, b, n_users, variant) for r in range(world)])
                ids_i = np.concatenate([synth.batch_ids(seed, synth.TID_ITEM_IDS, step * world + r, b, n_items, variant) for r in range(world)])
                masks = tuple([(t.acts[l + 1] > 0).cpu().numpy() for l in range(t.n_layers - 1)] for t in (tr.user_tower, tr.item_tower))
                allm = [None] * world
                dist.all_gather_object(allm, masks, group=og)
                gm = tuple([np.concatenate([allm[r][t][l] for r in range(world)]) for l in range(len(tower_dims) - 1)] for t in (0, 1))
                ids_c = np.concatenate([cats(step, r) for r in range(world)]) if nb else None
                fb = tt.forward_backward(ref, ids_u, ids_i, temperature=0.1, l2=0.0, relu_masks=gm, category_ids=ids_c)
                mine = fb["per_row"][rank * b:(rank + 1) * b].sum()
                assert abs(loss - mine) <= 1e-4 * abs(mine), (loss, mine)
                subs.append((ids_u, ids_i, fb, ids_c))
            for r in range(world if negatives == "local" else 0):
                ids_u = synth.batch_ids(seed, synth.TID_USER_IDS, step * world + r, b, n_users, variant)
                ids_i = synth.batch_ids(seed, synth.TID_ITEM_IDS, step * world + r, b, n_items, variant)
                if r == rank:
                    assert np.array_equal(u.cpu().numpy(), ids_u) and np.array_equal(i.cpu().numpy(), ids_i)
                    masks = tuple([(t.acts[l + 1] > 0).cpu().numpy() for l in range(t.n_layers - 1)]
                                  for t in (tr.user_tower, tr.item_tower))
                allm = [None] * world
                dist.all_gather_object(allm, masks if r == rank else None, group=og)
                fb = tt.forward_backward(ref, ids_u, ids_i, temperature=0.1, l2=0.0, relu_masks=allm[r], category_ids=cats(step, r))
                if r == rank:
                    assert abs(loss - fb["loss"]) <= 1e-4 * abs(fb["loss"]), (loss, fb["loss"])
                subs.append((ids_u, ids_i, fb, cats(step, r)))
            gu = np.concatenate([s[2]["due"] for s in subs]); iu = np.concatenate([s[0] for s in subs])
            gi = np.concatenate([s[2]["die"] for s in subs]); ii = np.concatenate([s[1] for s in subs])
            ic = np.concatenate([s[3] for s in subs]) if nb else None
            if opt == "sgd":
                tt.sparse_sgd(ref.user_table, iu, gu, 0.001); tt.sparse_sgd(ref.item_table, ii, gi, 0.001)
                if nb:
                    tt.sparse_sgd(ref.cat_table, ic, gi, 0.001)
            else:
                tt.sparse_adagrad(ref.user_table, ref.user_accum, iu, gu, 0.001)
                tt.sparse_adagrad(ref.item_table, ref.item_accum, ii, gi, 0.001)
                if nb:
                    tt.sparse_adagrad(ref.cat_table, ref.cat_accum, ic, gi, 0.001)
            if nb:       # replicated table: every rank holds all rows (summed with the dense all-reduce)
                assert np.abs(tr.cat_table.cpu().numpy() - ref.cat_table).max() <= 3e-6
            for tw, kw_, kb_ in ((ref.user_tower, "udw", "udb"), (ref.item_tower, "idw", "idb")):
                for l in range(len(tw.weights)):
                    gw = sum(s[2][kw_][l] for s in subs) + 2e-6 * tw.weights[l]
                    gb = sum(s[2][kb_][l] for s in subs)
                    if opt == "sgd":
                        tt.dense_sgd(tw.weights[l], gw, 0.001); tt.dense_sgd(tw.biases[l], gb, 0.001)
                    else:
                        tt.dense_adagrad(tw.weights[l], tw.w_accum[l], gw, 0.001); tt.dense_adagrad(tw.biases[l], tw.b_accum[l], gb, 0.001)
            for shard, full in ((tr.user_table, ref.user_table), (tr.item_table, ref.item_table)):
                mine = full[rank::world]
                got = shard.cpu().numpy()
                assert got.shape == mine.shape
                assert np.abs(got - mine).max() <= 3e-6, np.abs(got - mine).max()
            for tower, rt in ((tr.user_tower, ref.user_tower), (tr.item_tower, ref.item_tower)):
                for l in range(len(tower_dims)):
                    assert np.abs(tower.w[l].cpu().numpy() - rt.weights[l]).max() <= 1e-5
                    assert np.abs(tower.b[l].cpu().numpy() - rt.biases[l]).max() <= 1e-5
        # forward-only paths on the updated state: validation loss, the all-gathered item corpus, corpus ranks
        ids_u = [synth.batch_ids(seed, synth.TID_USER_IDS, 1 * world + r, b, n_users, variant) for r in range(world)]
        ids_i = [synth.batch_ids(seed, synth.TID_ITEM_IDS, 1 * world + r, b, n_items, variant) for r in range(world)]
        val = tr.evaluate(u, i, category_ids=dc).item()
        if negatives == "local":
            want = tt.forward_backward(ref, ids_u[rank], ids_i[rank], temperature=0.1, category_ids=cats(1, rank))["loss"]
        else:
            fbv = tt.forward_backward(ref, np.concatenate(ids_u), np.concatenate(ids_i), temperature=0.1,
                                      category_ids=np.concatenate([cats(1, r) for r in range(world)]) if nb else None)
            want = fbv["per_row"][rank * b:(rank + 1) * b].sum()
        assert abs(val - want) <= 1e-4 * abs(want), (val, want)
        item_cat = (np.arange(n_items) * 7) % nb if nb else None
        corpus = tr.item_corpus_embeddings(None if item_cat is None else torch.from_numpy(item_cat).to(dev))
        rows = ref.item_table if item_cat is None else ref.item_table + ref.cat_table[item_cat]
        want_c = tt.tower_fwd(rows, ref.item_tower.weights, ref.item_tower.biases)[-1]
        assert corpus.shape == want_c.shape
        assert np.abs(corpus.cpu().numpy() - want_c).max() <= 1e-4 * np.abs(want_c).max()
        from two_tower_amazon_recommender_amd.metrics import FactorizedTopK
        metric = FactorizedTopK(ks=(1, 10, 100), temperature=0.1)
        ranks = tr.evaluate_topk(u, i, metric, corpus).cpu().numpy()
        qv = tt.tower_fwd(ref.user_table[ids_u[rank]], ref.user_tower.weights, ref.user_tower.biases)[-1]
        lo, hi = tt.retrieval_rank_bounds(qv, want_c, ids_i[rank], temperature=0.1)
        assert (ranks >= lo).all() and (ranks <= hi).all()
        # replicas of the dense parameters stay bit-identical across ranks
        flat = [None] * world
        dist.all_gather_object(flat, tr.dense_flat.cpu().numpy(), group=og)
        assert np.array_equal(flat[0], flat[1])
        ret[rank] = "ok"
    except Exception:                                             # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def _worker_options(rank, world, port, negatives, ret, real=False):
    """sample_weight + candidate_sampling_probability + accidental-hit removal through the SHARDED step (weights stay
    with the rank's queries; probabilities and candidate ids are all-gathered with the candidates when negatives are
    global) against the f64 oracle on the global batch."""
    try:
        dev, og = _init(rank, world, port, real)
        from oracle import synth, two_tower as tt
        from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer
        from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig
        n_users, n_items, dim, tower_dims, b, seed = 3001, 300, 64, [64], 512, 43      # 300 items: many accidental hits
        cfg = TwoTowerConfig(n_users=n_users, n_items=n_items, embedding_dim=dim, tower_dims=tower_dims, temperature=0.1,
                             l2_regularization=0.0, learning_rate=0.01, optimizer="sgd", batch_size=b)
        tr = ShardedTwoTowerTrainer(cfg, dev, seed=seed, negatives=negatives, capacity_factor=4.0)
        ref = tt.synthetic_state(seed, n_users, n_items, dim, tower_dims, dtype=np.float64, optimizer="sgd")
        w_all = synth.uniform_f32(seed, 77, world * b, 0.5, 1.0)
        p_all = synth.uniform_f32(seed, 78, world * b, 0.001, 0.5)
        u, i = tr.synthetic_batch(seed, 0, "Z")
        sl = slice(rank * b, (rank + 1) * b)
        # set_negatives(): the OTHER slab shape on the same trainer (bench_dist's second region), forward only, then back
        other = "local" if negatives == "global" else "global"
        tr.set_negatives(other)
        ev = tr.evaluate(u, i).item()
        eu = [synth.batch_ids(seed, synth.TID_USER_IDS, r, b, n_users, "Z") for r in range(world)]
        ei = [synth.batch_ids(seed, synth.TID_ITEM_IDS, r, b, n_items, "Z") for r in range(world)]
        if other == "global":
            want_ev = tt.forward_backward(ref, np.concatenate(eu), np.concatenate(ei), temperature=0.1)["per_row"][sl].sum()
        else:
            want_ev = tt.forward_backward(ref, eu[rank], ei[rank], temperature=0.1)["loss"]
        assert abs(ev - want_ev) <= 1e-4 * abs(want_ev), (ev, want_ev)
        tr.set_negatives(negatives)
        loss = tr.step(u, i, sample_weight=torch.from_numpy(w_all[sl].copy()).to(dev),
                       candidate_sampling_probability=torch.from_numpy(p_all[sl].copy()).to(dev), candidate_ids=i).item()
        tr.check_ids()
        ids_u = [synth.batch_ids(seed, synth.TID_USER_IDS, r, b, n_users, "Z") for r in range(world)]
        ids_i = [synth.batch_ids(seed, synth.TID_ITEM_IDS, r, b, n_items, "Z") for r in range(world)]
        assert np.array_equal(i.cpu().numpy(), ids_i[rank])
        if negatives == "global":
            au, ai = np.concatenate(ids_u), np.concatenate(ids_i)
            fb = tt.forward_backward(ref, au, ai, temperature=0.1, sample_weight=w_all, candidate_sampling_probability=p_all,
                                     candidate_ids=ai, remove_accidental_hits=True)
            want = fb["per_row"][sl].sum()
            gu, gi, iu, ii = fb["due"], fb["die"], au, ai
        else:
            fbs = [tt.forward_backward(ref, ids_u[r], ids_i[r], temperature=0.1, sample_weight=w_all[r * b:(r + 1) * b],
                                       candidate_sampling_probability=p_all[r * b:(r + 1) * b], candidate_ids=ids_i[r],
                                       remove_accidental_hits=True) for r in range(world)]
            want = fbs[rank]["loss"]
            gu, gi = np.concatenate([f["due"] for f in fbs]), np.concatenate([f["die"] for f in fbs])
            iu, ii = np.concatenate(ids_u), np.concatenate(ids_i)
        assert abs(loss - want) <= 1e-4 * abs(want), (loss, want)
        tt.sparse_sgd(ref.user_table, iu, gu, 0.01); tt.sparse_sgd(ref.item_table, ii, gi, 0.01)
        for shard, full in ((tr.user_table, ref.user_table), (tr.item_table, ref.item_table)):
            err = np.abs(shard.cpu().numpy() - full[rank::world]).max()
            assert err <= 3e-6, err
        # a wrong-length option vector is refused before anything is enqueued
        with pytest.raises(ValueError):
            tr.step(u, i, sample_weight=torch.ones(b - 1, device=dev))
        ret[rank] = "ok"
    except Exception:                                             # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _spawn(fn, args, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(fn, args=(world, _free_port()) + args + (ret,), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"


@pytest.mark.parametrize("negatives", ["global", "local"])
def test_sharded_step_sample_weight_logq_and_accidental_hits_two_ranks(negatives):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _spawn(_worker_options, (negatives,))


def _worker_real(rank, world, port, opt, variant, negatives, nb, ret):
    _worker(rank, world, port, opt, variant, negatives, nb, ret, real=True)


def _worker_options_real(rank, world, port, negatives, ret):
    _worker_options(rank, world, port, negatives, ret, real=True)


@pytest.mark.parametrize("opt,variant,negatives,nb", [("sgd", "U", "local", 0), ("adagrad", "Z", "global", 0),
                                                      ("sgd", "Z", "global", 30)])
def test_sharded_trainer_real_rccl_one_gpu_per_rank(opt, variant, negatives, nb):
    """The same 2-step comparison against the f64 oracle with the product's OWN collectives: a real "nccl" (RCCL) group,
    one GPU per rank, inline synchronous C1/C2/C6, asynchronous C3 beside the dw GEMMs, the side-stream sort plans.
    Needs >= 2 visible GPUs: skipped on the 1-GPU test box, runs the first time a multi-GPU lease is available."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (multi-GPU RCCL correctness is UNVERIFIED until this runs)")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_real, args=(2, _free_port(), opt, variant, negatives, nb, ret), nprocs=2, join=True)
    for r in range(2):
        assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"
    mp.spawn(_worker_options_real, args=(2, _free_port(), negatives, ret), nprocs=2, join=True)
    for r in range(2):
        assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"


@pytest.mark.parametrize("opt,variant,negatives,nb", [("sgd", "U", "local", 0), ("adagrad", "Z", "local", 30),
                                                      ("sgd", "Z", "global", 30)])
def test_sharded_trainer_two_ranks_on_one_gpu(opt, variant, negatives, nb):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), opt, variant, negatives, nb, ret), nprocs=2, join=True)
    for r in range(2):
        assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"
