"""world_size-2, 3 and 8 gloo tests of the row-sharded embedding exchange on CPU (the N>1 path of
SURVEY.md §8e).  The exchange code under test is the product's (device-agnostic torch +
torch.distributed); the three row kernels are replaced by a NumPy-oracle backend DEFINED HERE (the
product's only backend is HIP).  Checks: looked-up rows, post-step shards (SGD and Adagrad, duplicates
inside and across ranks), padding, capacity overflow and out-of-range reporting."""
import os
import socket
import sys
import pathlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = pathlib.Path(__file__).resolve().parents[1]


class OracleRowBackend:
    """Test-only stand-in for HipRowBackend: NumPy restatements of the row kernels (route = stable partition by
    owner, gather, scatter_rows, sort+apply through the oracle's sparse optimizers)."""

    def route(self, ids_list, world, num_rows_list, offsets, cap, send_ids, pos_flats, flags):
        nt = len(ids_list)
        send = np.full(world * nt * cap, -1, dtype=np.int64)
        for t in range(nt):
            i = ids_list[t].numpy()
            pos = np.full(len(i), -1, dtype=np.int64)
            fill = [0] * world
            for p, v in enumerate(i):
                if v < 0 or v >= num_rows_list[t]:
                    flags[0] = 1
                    continue
                o = int(v % world)
                if fill[o] < cap:
                    slot = (o * nt + t) * cap + fill[o]
                    send[slot] = v // world + offsets[t]
                    pos[p] = slot
                else:
                    flags[1] = 1
                fill[o] += 1
            pos_flats[t].copy_(torch.from_numpy(pos))
        send_ids.copy_(torch.from_numpy(send))

    def gather(self, table, ids, out, oob_flag):
        t, i = table.numpy(), ids.numpy()
        ok = (i >= 0) & (i < t.shape[0])
        res = np.zeros((len(i), t.shape[1]), dtype=np.float32)
        res[ok] = t[i[ok]]
        out.copy_(torch.from_numpy(res))
        if oob_flag is not None and ((~ok) & (i != -1)).any():
            oob_flag.fill_(1)

    def scatter_rows(self, src, idx, dst):
        i = idx.numpy()
        ok = i >= 0
        dst.numpy()[i[ok]] = src.numpy()[ok]

    def plan(self, ids, num_rows, key=None):
        pass

    def prefetch_stream(self):
        import contextlib
        return contextlib.nullcontext()

    def join_prefetch(self):
        pass

    def apply(self, opt, table, accum, ids, grads, lr, eps, key=None):
        from oracle import two_tower as tt
        i = ids.numpy()
        keep = i >= 0
        if opt == "sgd":
            tt.sparse_sgd(table.numpy(), i[keep], grads.numpy()[keep], lr)
        else:
            tt.sparse_adagrad(table.numpy(), accum.numpy(), i[keep], grads.numpy()[keep], lr, eps)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, ret):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import synth, two_tower as tt
        from two_tower_amazon_recommender_amd.sharded import ShardedEmbedding, shard_rows
        rows, dim, batch, opt, variant, capf = case
        full = synth.embedding_table(3, 1, rows, dim)                       # the global table (oracle side)
        n_local = shard_rows(rows, world, rank)
        shard = torch.from_numpy(full[rank::world].copy())
        assert shard.shape[0] == n_local
        accum = torch.full_like(shard, 0.1) if opt == "adagrad" else None
        emb = ShardedEmbedding(rows, dim, batch, torch.device("cpu"), capacity_factor=capf, backend=OracleRowBackend(),
                               table=shard, accum=accum)
        ref_table = full.copy()
        ref_acc = np.full_like(full, np.float32(0.1))
        for step in range(2):
            all_ids = synth.batch_ids(3, 3, step, world * batch, rows, variant)
            all_g = synth.uniform_f32(3, 9 + step, world * batch * dim, -1.0, 2.0).reshape(world * batch, dim)
            ids = all_ids[rank * batch:(rank + 1) * batch]
            g = all_g[rank * batch:(rank + 1) * batch]
            out = torch.empty(batch, dim)
            emb.lookup(torch.from_numpy(ids), out)
            assert np.array_equal(out.numpy(), ref_table[ids]), "looked-up rows differ"
            emb.apply_gradients(torch.from_numpy(g), opt, 0.01, 1e-7)
            emb.check()
            # oracle on the GLOBAL batch; summation order differs (per-rank partial sums first), so compare with a
            # tolerance of a few f32 ulps of the gradient sum
            if opt == "sgd":
                tt.sparse_sgd(ref_table, all_ids, all_g, 0.01)
            else:
                tt.sparse_adagrad(ref_table, ref_acc, all_ids, all_g, 0.01, 1e-7)
            got = emb.table.numpy()[:n_local]
            assert np.allclose(got, ref_table[rank::world], rtol=0, atol=2e-6), np.abs(got - ref_table[rank::world]).max()
            # after the comparison re-sync the oracle to the sharded state so errors do not compound
            gathered = [None] * world
            dist.all_gather_object(gathered, got.copy())
            for r in range(world):
                ref_table[r::world] = gathered[r]
            if opt == "adagrad":
                gacc = [None] * world
                dist.all_gather_object(gacc, emb.accum.numpy()[:n_local].copy())
                for r in range(world):
                    ref_acc[r::world] = gacc[r]
        ret[rank] = "ok"
    except Exception as e:                                                    # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def _run(world, case):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), case, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"


@pytest.mark.parametrize("case", [
    (1000, 32, 256, "sgd", "U", 2.0),
    (1000, 32, 256, "adagrad", "Z", 2.0),       # heavy duplicates inside and across ranks
    (37, 8, 64, "sgd", "U", 4.0),               # tiny table: every id duplicated many times; ragged shards (cap = batch)
])
def test_sharded_embedding_world2(case):
    _run(2, case)


def test_sharded_embedding_world3_ragged():
    _run(3, (1001, 16, 96, "adagrad", "Z", 3.0))


def _worker_flags(rank, world, port, ret):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from two_tower_amazon_recommender_amd.sharded import ShardedEmbedding
        emb = ShardedEmbedding(1000, 8, 128, torch.device("cpu"), capacity_factor=1.0, backend=OracleRowBackend(),
                               table=torch.zeros(500, 8))
        assert emb.cap == 64
        out = torch.empty(128, 8)
        ids = torch.arange(128, dtype=torch.int64) * 2            # all even: 128 positions for owner 0 > cap 64
        emb.lookup(ids, out)
        try:
            emb.check()
            ret[rank] = "no overflow error"
        except RuntimeError as e:
            ret[rank] = "ok" if "overflow" in str(e) else str(e)
        ids = torch.arange(128, dtype=torch.int64)
        ids[5] = 1000                                             # out of range
        emb.lookup(ids, out)
        try:
            emb.check()
            ret[rank] = "no oob error"
        except IndexError:
            assert not out[5].any()
        # poll(): the asynchronous form — the copy started by one poll is examined by the next, the error names the step
        emb.poll(10)                                              # clean flags: nothing to report
        emb.lookup(ids, out)                                      # sets the out-of-range flag again
        emb.poll(60)                                              # looks at the copy of step 10 (clean), copies again
        try:
            emb.poll(110)
            ret[rank] = "poll reported nothing"
        except IndexError as e:
            if "step 60" not in str(e):
                ret[rank] = f"poll error does not name the step: {e}"
        emb.poll(160)
        emb.poll(210)                                             # flags were cleared by the raise: quiet again
    except Exception:                                             # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_overflow_and_out_of_range_are_reported():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_flags, args=(2, _free_port(), ret), nprocs=2, join=True)
    for r in range(2):
        assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"


def _worker_combined(rank, world, port, ret):
    """Two tables behind ONE set of exchange buffers, as ShardedTwoTowerTrainer.step drives them (async collectives):
    ragged shards (700 and 333 rows over 2 and 3 ranks), heavy duplicates, SGD and Adagrad."""
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import synth, two_tower as tt
        from two_tower_amazon_recommender_amd.sharded import ShardedTables, shard_rows
        dim, batch = 16, 128
        # world 8 (r04, VERDICT r03 item 4: the world size the machine has): a THIRD table of 30 rows beside the two (the
        # shape of a sharded category table: 30 % 8 != 0, every id hundreds of times), and room for the power-law ids' hot
        # owner (id 0 of the 30-row table alone draws ~43 % of a rank's positions: capacity = the whole batch)
        rows_list = [700, 333] + ([30] if world >= 8 else [])
        nt = len(rows_list)
        for opt in ("sgd", "adagrad"):
            fulls = [synth.embedding_table(5, 1 + t, r, dim) for t, r in enumerate(rows_list)]
            accs = [np.full_like(f, np.float32(0.1)) for f in fulls]
            em = ShardedTables(rows_list, dim, batch, torch.device("cpu"), capacity_factor=3.0 if world < 8 else float(world),
                               backend=OracleRowBackend())
            assert em.offsets[:2] == [0, (700 + world - 1) // world] and em.table.shape[0] == sum(em.rows_cap)
            for t in range(nt):
                em.shard(t).copy_(torch.from_numpy(fulls[t][rank::world].copy()))
                assert em.shard(t).shape[0] == shard_rows(fulls[t].shape[0], world, rank)
            if opt == "adagrad":
                em.accum = torch.full_like(em.table, 0.1)
            n_steps = 3
            all_ids = [[synth.batch_ids(5, 3 + t, step, world * batch, fulls[t].shape[0], "Z") for t in range(nt)]
                       for step in range(n_steps)]
            sl = slice(rank * batch, (rank + 1) * batch)
            mine = [[torch.from_numpy(all_ids[step][t][sl]) for t in range(nt)] for step in range(n_steps)]
            for step in range(n_steps):
                ids = all_ids[step]
                grads = [synth.uniform_f32(5, 9 + nt * step + t, world * batch * dim, -1.0, 2.0).reshape(world * batch, dim)
                         for t in range(nt)]
                out = torch.empty(nt * batch, dim)
                cur_before = em._cur
                em.lookup_start(mine[step])
                # steps 1.. were prefetched during the previous step: the other id-buffer set is adopted
                assert (em._cur != cur_before) == (step > 0)
                em.lookup_rows()
                em.lookup_finish(out)
                if step + 1 < n_steps:
                    em.lookup_prefetch(mine[step + 1], exchange=(step % 2 == 0))   # next step's route (+ id exchange), issued mid-step
                for t in range(nt):
                    assert np.array_equal(out.numpy()[t * batch:(t + 1) * batch], fulls[t][ids[t][sl]])
                em.grads_start(torch.from_numpy(np.concatenate([grads[t][sl] for t in range(nt)])))
                em.grads_finish(opt, 0.01)
                em.check()
                for t in range(nt):
                    if opt == "sgd":
                        tt.sparse_sgd(fulls[t], ids[t], grads[t], 0.01)
                    else:
                        tt.sparse_adagrad(fulls[t], accs[t], ids[t], grads[t], 0.01, 1e-7)
                    got = em.shard(t).numpy()
                    assert np.allclose(got, fulls[t][rank::world], rtol=0, atol=2e-6), (opt, t, np.abs(got - fulls[t][rank::world]).max())
                    # re-sync the oracle to the sharded state so last-bit differences do not compound
                    gathered = [None] * world
                    dist.all_gather_object(gathered, got.copy())
                    for r in range(world):
                        fulls[t][r::world] = gathered[r]
                    if opt == "adagrad":
                        gacc = [None] * world
                        dist.all_gather_object(gacc, em.accum_shard(t).numpy().copy())
                        for r in range(world):
                            accs[t][r::world] = gacc[r]
            # a prefetch for ids that are NOT the next call's: dropped, the lookup routes afresh
            em.lookup_prefetch(mine[0])
            out = torch.empty(nt * batch, dim)
            em.lookup(mine[2], out)
            for t in range(nt):
                assert np.array_equal(out.numpy()[t * batch:(t + 1) * batch], fulls[t][all_ids[2][t][sl]])
            em.check()
        if world >= 8:
            # the overflow case at 8 owners: capacity_factor 2.0 reserves 64 positions per owner and table (2 x 128 / 8, rounded
            # up to 64); a batch whose 30-row-table ids are ALL id 8 sends every one of a rank's 128 positions to owner 0 -
            # flagged, on every rank, by the next check() (and the rows of the tables that did fit are still the right ones)
            em2 = ShardedTables(rows_list, dim, batch, torch.device("cpu"), capacity_factor=2.0, backend=OracleRowBackend())
            assert em2.cap == 64
            for t in range(nt):
                em2.shard(t).copy_(torch.from_numpy(fulls[t][rank::world].copy()))
            out = torch.zeros(nt * batch, dim)
            hot = [mine[0][0], mine[0][1], torch.full((batch,), 8, dtype=torch.int64)]
            em2.lookup(hot, out)
            try:
                em2.check()
                raise AssertionError("no overflow reported at world 8 with capacity_factor 2.0")
            except RuntimeError as e:
                assert "overflow" in str(e), e
            ok = 0
            for p in range(batch):                                # table 0 (700 rows): no owner draws more than 32 of 128
                got = out.numpy()[p]
                ok += int(np.array_equal(got, fulls[0][all_ids[0][0][sl][p]]))
            assert ok == batch, ok
        ret[rank] = "ok"
    except Exception:                                             # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_two_tables_one_exchange(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_combined, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"


def _worker_cand_ids(rank, world, port, ret):
    """The all-gathers that build the global-negatives candidate set, as ShardedTwoTowerTrainer issues them (C4 for the
    embeddings, _cand_ids / _cand_prob for their ids and sampling probabilities): gloo, no GPU."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from types import SimpleNamespace
        from two_tower_amazon_recommender_amd.sharded import ShardedTwoTowerTrainer as T
        b, d = 5, 4
        me = SimpleNamespace(negatives="global", collectives=True, world=world, rank=rank, dev=torch.device("cpu"),
                             group=dist.group.WORLD, cfg=SimpleNamespace(batch_size=b))
        ids = torch.arange(b, dtype=torch.int64) * 7 + 1000 * rank                  # rank r's candidate ids
        prob = (torch.arange(b, dtype=torch.float32) + 1) / (10.0 * (rank + 1))
        c = ids.to(torch.float32)[:, None].repeat(1, d)                             # an embedding row that names its id
        id_all = T._cand_ids(me, ids)
        p_all = T._cand_prob(me, prob)
        c_all = torch.empty(world * b, d)
        dist.all_gather_into_tensor(c_all, c, group=me.group)                       # C4, exactly as in step() / evaluate()
        # rank-major order, the same for all three; this rank's own candidates (the positives) start at diag_offset = rank * b
        assert id_all.shape == (world * b,) and p_all.shape == (world * b,)
        for r in range(world):
            assert torch.equal(id_all[r * b:(r + 1) * b], torch.arange(b, dtype=torch.int64) * 7 + 1000 * r)
            assert torch.equal(p_all[r * b:(r + 1) * b], (torch.arange(b, dtype=torch.float32) + 1) / (10.0 * (r + 1)))
        assert torch.equal(c_all[:, 0].to(torch.int64), id_all)                     # row i of c_all IS candidate id_all[i]
        off = rank * b
        assert torch.equal(id_all[off:off + b], ids) and torch.equal(p_all[off:off + b], prob)
        # local negatives / no collectives: the rank's own tensors, untouched
        me.negatives = "local"
        assert T._cand_ids(me, ids) is ids and T._cand_prob(me, prob) is prob
        me.negatives, me.collectives = "global", False
        assert T._cand_ids(me, ids) is ids and T._cand_ids(me, None) is None
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback
        ret[rank] = traceback.format_exc() + repr(e)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_global_negatives_candidate_ids_follow_the_candidate_rows(world):
    """id_all / p_all must be in the order of c_all (rank-major) and a rank's own block must start at its
    diag_offset = rank * batch - otherwise accidental-hit removal and the log-Q correction act on the wrong columns."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_cand_ids, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"
