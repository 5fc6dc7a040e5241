#!/bin/bash
# same-box A/B of the optimizer launch: variant libs given as arguments (names under scratch/variants, or "main");
# rocprof kernel stats of 220 bench steps + per-workgroup phase stamps of the matching -DTT_SORT_STAMPS build
out=gpurun_out/r03; mkdir -p $out
k=0
for v in "$@"; do
  k=$((k+1))
  if [ "$v" = main ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$PWD/scratch/variants/$v.so; fi
  bash scratch/prof.sh ab${k}_$v --steps 200 --warmup 20 > /dev/null 2>&1
  python - > $out/ab_opt_${k}_$v.txt <<PY
import csv, json
rows = list(csv.DictReader(open('gpurun_out/prof_ab${k}_$v/trace_kernel_stats.csv')))
for r in rows:
    n = r['Name']
    if any(s in n for s in ('optimizer_ids', 'tower_fwd2', 'gemm_kernel<true, false, false, 1', 'gemm_bwd_kernel<0>')):
        print(f"{n[28:70]:42s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:7.2f}  max {float(r['MaxNs'])/1e3:7.2f}")
try:
    print('ms_per_step', json.loads(open('gpurun_out/prof_ab${k}_$v/bench.json').read().strip().split('\n')[-1])['ms_per_step'])
except Exception as e:
    print('bench line unreadable', e)
PY
  rm -f gpurun_out/prof_ab${k}_$v/trace_kernel_trace.csv
  s=${v}_stamps; [ "$v" = main ] && s=stamps
  if [ -f scratch/variants/$s.so ]; then TT_LIB_PATH=$PWD/scratch/variants/$s.so timeout -k 10 200 python scratch/opt_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $out/ab_opt_${k}_$v.txt; fi
done
unset TT_LIB_PATH
for f in $out/ab_opt_[0-9]*; do echo "== $f"; cat $f; done
