"""Scorer training entry at the per-GPU SLAB shapes of the 8-GPU configurations:
nq x nc x dim = 2048 x 16384 x 128 (cfg4 at N = 8) and 4096 x 32768 x 256 (cfg5 at N = 8), next to the square.
Live kernel durations through tt_profile_* (dispatch timestamps); run under rocprofv3 --kernel-trace --stats for the committed
summary.  usage: r04_slab.py [nqxncxdim ...] [--prec f32|bf16x3|both] [--iters n]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_amazon_recommender_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
PEAK_F32 = 157.3e12


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    prec_arg = "f32"
    iters = 40
    for i, a in enumerate(sys.argv):
        if a == "--prec":
            prec_arg = sys.argv[i + 1]
        if a == "--iters":
            iters = int(sys.argv[i + 1])
    args = [a for a in args if "x" in a]
    shapes = [tuple(int(v) for v in a.split("x")) for a in args] or [(8192, 8192, 128), (2048, 16384, 128), (4096, 32768, 256)]
    precs = ("f32", "bf16x3") if prec_arg == "both" else (prec_arg,)
    # device warm-up (clocks): 0.4 s of the square scorer, as bench.py's pre-spin does
    import time
    wq = torch.empty(8192, 128, device=dev); ops.fill_uniform_(wq, 2, 1, -0.3, 0.6)
    wws = torch.empty(ops.retrieval_workspace_bytes(8192, 8192, 128), dtype=torch.uint8, device=dev)
    wl = torch.empty(8192, device=dev); wp = torch.empty(8192, device=dev); wloss = torch.empty(1, device=dev)
    wdq = torch.empty(8192, 128, device=dev); wdc = torch.empty(8192, 128, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        for _ in range(8):
            ops.retrieval_fwd_bwd(wq, wq, 10.0, wws, wl, wp, wloss, wdq, wdc)
        torch.cuda.synchronize()
    del wws
    for nq, nc, d in shapes:
        q = torch.empty(nq, d, device=dev); c = torch.empty(nc, d, device=dev)
        ops.fill_uniform_(q, 2, 1, -0.3, 0.6); ops.fill_uniform_(c, 2, 2, -0.3, 0.6)
        ws = torch.empty(ops.retrieval_workspace_bytes(nq, nc, d), dtype=torch.uint8, device=dev)
        lse = torch.empty(nq, device=dev); pr = torch.empty(nq, device=dev); loss = torch.empty(1, device=dev)
        dq = torch.empty(nq, d, device=dev); dc = torch.empty(nc, d, device=dev)
        off = (nc - nq) // 2 // 32 * 32 if nc > nq else 0          # the slab's diagonal sits somewhere inside the candidate range
        for prec in precs:
            fn = lambda: ops.retrieval_fwd_bwd(q, c, 10.0, ws, lse, pr, loss, dq, dc, diag_offset=off, precision=prec)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            _lib.profile_enable("score_fused,score_bwd,score_aux", 4 * iters)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            f = _lib.profile_read("score_fused", 4 * iters)[0]
            bw = _lib.profile_read("score_bwd", 4 * iters)[0]
            aux = _lib.profile_read("score_aux", 4 * iters)[0]
            _lib.profile_enable("")
            fu, bu = sum(f) / len(f) * 1e3, sum(bw) / len(bw) * 1e3
            comb = sum(aux[0::2]) / max(1, len(aux[0::2])) * 1e3
            red = sum(aux[1::2]) / max(1, len(aux[1::2])) * 1e3
            tot = fu + bu + comb + red
            flops = 6.0 * nq * nc * d
            print(json.dumps({"nq": nq, "nc": nc, "dim": d, "prec": prec, "pass1_us": round(fu, 2), "pass2_us": round(bu, 2),
                              "combine_us": round(comb, 2), "reduce_us": round(red, 2), "kernels_us": round(tot, 2),
                              "wall_us_per_call": round(e0.elapsed_time(e1) / iters * 1e3, 2),
                              "frac_f32_peak_all4": round(flops / (tot * 1e-6) / PEAK_F32, 4),
                              "frac_pass1": round(4.0 * nq * nc * d / (fu * 1e-6) / PEAK_F32, 4),
                              "frac_pass2": round(2.0 * nq * nc * d / (bu * 1e-6) / PEAK_F32, 4),
                              "brackets": os.environ.get("TT_PROF_BRACKETS", "0"), "loss": loss.item()}), flush=True)


if __name__ == "__main__":
    main()
