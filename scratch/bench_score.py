"""Scorer microbench: tt_retrieval_fwd_bwd in both precisions at cfg3/cfg4/cfg5 shapes (hipEvent, back-to-back)."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from two_tower_amazon_recommender_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def main():
    shapes = ((8192, 128), (16384, 128), (8192, 256), (32768, 256))
    if len(sys.argv) > 1:
        shapes = tuple(tuple(int(v) for v in a.split("x")) for a in sys.argv[1:])
    for b, d in shapes:
        q = torch.empty(b, d, device=dev); c = torch.empty(b, d, device=dev)
        ops.fill_uniform_(q, 2, 1, -0.3, 0.6); ops.fill_uniform_(c, 2, 2, -0.3, 0.6)
        ws = torch.empty(ops.retrieval_workspace_bytes(b, b, d), dtype=torch.uint8, device=dev)
        lse = torch.empty(b, device=dev); pr = torch.empty(b, device=dev); loss = torch.empty(1, device=dev)
        dq = torch.empty(b, d, device=dev); dc = torch.empty(b, d, device=dev)
        out = {"batch": b, "dim": d}
        for prec in ("f32", "bf16x3"):
            fn = lambda: ops.retrieval_fwd_bwd(q, c, 10.0, ws, lse, pr, loss, dq, dc, precision=prec)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            _lib.profile_enable("score_fused,score_bwd", 64)
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            f = _lib.profile_read("score_fused", 64)[0]; bw = _lib.profile_read("score_bwd", 64)[0]
            _lib.profile_enable("")
            out[prec] = {"fused_us": sum(f) / len(f) * 1e3, "bwd_us": sum(bw) / len(bw) * 1e3, "loss": loss.item(),
                         "dq_abs_max": dq.abs().max().item()}
            # validation forward and the rank pass (pure GEMM1), hipEvents over back-to-back calls
            pos = torch.arange(b, device=dev)
            rk = torch.empty(b, dtype=torch.int32, device=dev)
            for name, g in (("fwd_us", lambda: ops.retrieval_fwd(q, c, 10.0, ws, lse, pr, loss, precision=prec)),
                            ("rank_us", lambda: ops.retrieval_rank(q, c, 10.0, pos, workspace=ws, out=rk, precision=prec))):
                for _ in range(2):
                    g()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    g()
                e1.record()
                torch.cuda.synchronize()
                out[prec][name] = e0.elapsed_time(e1) / 10 * 1e3
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
