"""kernel stats of a rocprofv3 --stats run: python scratch/kstats.py <trace_kernel_stats.csv> [substring ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:] or ['tower_fwd2', 'gemm_kernel', 'gemm_bwd', 'optimizer', 'score_kernel', 'fused_combine', 'reduce_slabs']
for r in rows:
    n = r['Name']
    if any(k in n for k in keys):
        print(f"{n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
