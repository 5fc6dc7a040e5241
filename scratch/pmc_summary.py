import csv, sys, collections, glob
base = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + "/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    if "score" in k or "gather" in k or "sparse_apply" in k or "gemm" in k:
        print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()})
