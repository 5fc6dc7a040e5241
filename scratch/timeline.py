"""Per-step timeline of the main queue from a rocprofv3 kernel trace: average duration of every kernel of a step
and the idle gap that follows it.  usage: python scratch/timeline.py gpurun_out/prof_<tag>/trace_kernel_trace.csv"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
cnt = collections.Counter(r["Queue_Id"] for r in rows if "score_kernel" in r["Kernel_Name"])
q = cnt.most_common(1)[0][0]
rows = sorted((r for r in rows if r["Queue_Id"] == q), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:46] for r in rows]
# a step opens with the forward pass: the fused two-layer tower kernel (r03), else the layer-0 GEMM with the fused lookup
opener = "tower_fwd2_kernel" if any(n.startswith("tower_fwd2_kernel") for n in names) else "gemm_kernel<true, false, false, 1"
g = [i for i, n in enumerate(names) if n.startswith(opener)]
steps = [(g[k], g[k + 1]) for k in range(len(g) - 1)]
# the timed region's steps only: the usual kernel count, exact-f32 scorer, fused optimizer launch
steps = [(a, b) for a, b in steps if b - a <= 12 and any("score_kernel<128, 4, false, false, 4, 0>" in names[j] or
                                                          "score_kernel<128, 2, false, false, 4, 0>" in names[j] for j in range(a, b))
         and any(names[j].startswith("optimizer_") for j in range(a, b))]
mode = collections.Counter(b - a for a, b in steps).most_common(1)[0][0]
steps = [(a, b) for a, b in steps if b - a == mode][-150:]
agg, gaps, wall = collections.OrderedDict(), collections.OrderedDict(), 0
for a, b in steps:
    wall += int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    for j in range(a, b):
        key = (j - a, names[j])
        agg[key] = agg.get(key, 0) + int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])
        gaps[key] = gaps.get(key, 0) + int(rows[j + 1]["Start_Timestamp"]) - int(rows[j]["End_Timestamp"])
n = len(steps)
print(f"steps {n}  wall/step {wall / n / 1e3:.1f} us")
tk = tg = 0.0
for key in agg:
    print(f"{key[0]:2d} {key[1]:48s} {agg[key] / n / 1e3:8.2f} us   gap after {gaps[key] / n / 1e3:7.2f}")
    tk += agg[key] / n / 1e3
    tg += gaps[key] / n / 1e3
print(f"kernels {tk:.1f} us  gaps {tg:.1f} us")
