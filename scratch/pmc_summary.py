"""Summarise rocprofv3 --pmc passes (scratch/prof_pmc.sh) into profiles/<tag>_pmc.json.
HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE reports half the bytes of wide
(16 B/lane) coalesced reads (MI355X_MICROARCH.md §HBM), WRITE_SIZE is exact for 16-B/lane stores."""
import collections, csv, glob, json, sys
base, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + "/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
KEYS = {"score_fused": "score_kernel<128, 2, false>", "score_bwd": "score_kernel<128, 1, false>", "gather": "gather_kernel<4>",
        "sparse_apply": "sparse_apply_kernel<0>", "dense_fwd": "gemm_kernel<true, false, false>",
        "dense_bwd_dx": "gemm_kernel<true, true, false>", "dense_bwd_dw": "gemm_kernel<false, false, true>"}
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
