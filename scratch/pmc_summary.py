"""Summarise rocprofv3 --pmc passes (scratch/prof_pmc.sh) into profiles/<tag>_pmc.json.
HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced reads (MI355X_MICROARCH.md §HBM), WRITE_SIZE is exact for 16-B/lane stores."""
import collections, csv, glob, json, sys
base, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + "/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
# (score_bwd: the exact-f32 dc pass runs 8-wave workgroups since r02 - r03's summary looked for the 4-wave name and found none)
KEYS = {"score_fused": "score_kernel<128, 4, false, false, 4, 0>", "score_bwd": "score_kernel<128, 5, false, false, 8, 0>",
        "tower_fwd2": "tower_fwd2_kernel<",
        "score_fused_bf16x3": "score_kernel<128, 4, false, false, 8, 1>", "score_bwd_bf16x3": "score_kernel<128, 5, false, false, 8, 1>",
        "gather": "gather_kernel<4", "sparse_apply": "sparse_apply_kernel<0>", "sparse_plan": "part_sort_kernel<",
        "optimizer": "optimizer_ids_kernel<0",
        "dense_fwd_lookup": "gemm_kernel<true, false, false, 1", "dense_fwd": "gemm_kernel<true, false, false, 0",
        "dense_bwd_lookup": "gemm_bwd_kernel<256>", "dense_bwd": "gemm_bwd_kernel<0>",
        "fused_combine": "fused_combine_kernel", "reduce_slabs": "reduce_slabs_kernel", "dense_update": "dense_update_kernel<0>"}
res = {}
for tag, sub in KEYS.items():
    for k, d in agg.items():
        if k.startswith(sub):
            m = {c: sum(v) / len(v) for c, v in d.items()}
            e = {"kernel": k[:80], "launches": len(next(iter(d.values()))), "counters_avg": m}
            if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
                e["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_WAVE_CYCLES" in m:
                e["note"] = "SQ_VALU_MFMA_BUSY_CYCLES = MFMA issue cycles summed over SIMDs; SQ_WAVE_CYCLES in quad-cycles summed over waves"
            res[tag] = e
json.dump({"source": "rocprofv3 --kernel-trace --pmc (separate passes: FETCH_SIZE | WRITE_SIZE | SQ/GRBM), bench.py cfg3 sgd",
           "kernels": res}, open(out, "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in res.items()}))
