"""Tower GEMM launches exactly as the cfg3 train step issues them (both towers per launch): fwd L0, fwd L1, bwd L1
(dx + dw/db), bwd L0 — hipEvent time over back-to-back launches, per launch and summed, with the f32-MFMA fraction."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from two_tower_amazon_recommender_amd import ops  # noqa: E402
from two_tower_amazon_recommender_amd.trainer import TwoTowerConfig, TwoTowerTrainer, towers_forward, towers_backward  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    b, d, dims = 8192, 128, [256, 128]
    cfg = TwoTowerConfig(n_users=100_000, n_items=100_000, embedding_dim=d, tower_dims=dims, batch_size=b)
    tr = TwoTowerTrainer(cfg, dev, seed=3)
    u, i = tr.synthetic_batch(3, 0)
    tr.step(u, i)
    ut, it = tr.user_tower, tr.item_tower
    for tw in (ut, it):
        for z in tw.dz:
            z.normal_()
    none2 = (None, None)
    res = {}
    for l in range(2):
        hidden = l < 1
        res[f"fwd_L{l}"] = t(lambda: ops.dense_fwd2((ut.acts[l], it.acts[l]), (ut.w[l], it.w[l]), (ut.b[l], it.b[l]),
                                                    (ut.acts[l + 1], it.acts[l + 1]), relu=hidden, relu_bits=(ut.bits[l + 1], it.bits[l + 1])))
    for l in (1, 0):
        dxs = (ut.dz[l - 1], it.dz[l - 1]) if l > 0 else (ut.demb, it.demb)
        masks = (ut.acts[l], it.acts[l]) if l > 0 else none2
        bits = (ut.bits[l], it.bits[l]) if l > 0 else none2        # the train step's form: sign bits instead of the float mask
        res[f"bwd_L{l}"] = t(lambda: ops.dense_bwd2((ut.acts[l], it.acts[l]), (ut.w[l], it.w[l]), (ut.dz[l], it.dz[l]), dxs, none2,
                                                    (ut.dw_slabs[l], it.dw_slabs[l]), (ut.db_slabs[l], it.db_slabs[l]), dx_relu_bits=bits))
        if l > 0:
            res[f"bwd_L{l}_float_mask"] = t(lambda: ops.dense_bwd2((ut.acts[l], it.acts[l]), (ut.w[l], it.w[l]), (ut.dz[l], it.dz[l]), dxs, masks,
                                                                   (ut.dw_slabs[l], it.dw_slabs[l]), (ut.db_slabs[l], it.db_slabs[l])))
        res[f"bwd_L{l}_dx_only"] = t(lambda: ops.dense_bwd2((ut.acts[l], it.acts[l]), (ut.w[l], it.w[l]), (ut.dz[l], it.dz[l]), dxs,
                                                            none2, none2, none2, dx_relu_bits=bits))
        res[f"bwd_L{l}_dw_only"] = t(lambda: ops.dense_bwd2((ut.acts[l], it.acts[l]), (ut.w[l], it.w[l]), (ut.dz[l], it.dz[l]), none2,
                                                            none2, (ut.dw_slabs[l], it.dw_slabs[l]), (ut.db_slabs[l], it.db_slabs[l])))
    res["towers_fwd_all"] = t(lambda: towers_forward(ut, it))
    res["towers_bwd_all"] = t(lambda: towers_backward(ut, it))
    step_sum = res["fwd_L0"] + res["fwd_L1"] + res["bwd_L1"] + res["bwd_L0"]
    flops = 12 * 2.0 * b * 128 * 256
    res["sum_4_launches_us"] = step_sum
    res["frac_f32_mfma_peak"] = flops / (step_sum * 1e-6) / 157.3e12
    print(json.dumps(res))


if __name__ == "__main__":
    main()
