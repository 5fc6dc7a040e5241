#!/bin/bash
# cfg1 / cfg2: step time + per-kernel durations
out=gpurun_out/r03small; mkdir -p $out
for c in cfg1 cfg2; do
  timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --steps 2000 --warmup 200 | grep '^{' > $out/bench_$c.json
  bash scratch/prof.sh small_$c --config $c --steps 1000 --warmup 100 > /dev/null 2>&1
  python - <<PY
import csv, json
d = json.loads(open('$out/bench_$c.json').read().strip().splitlines()[-1])
print('$c', 'ms_per_step', d['ms_per_step'])
rows = list(csv.DictReader(open('gpurun_out/prof_small_$c/trace_kernel_stats.csv')))
tot = 0
for r in rows:
    per = float(r['TotalDurationNs']) / 1e3 / 1100
    if int(r['Calls']) >= 1000:
        tot += per
        print(f"  {r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:60]:60s} calls/step {int(r['Calls'])/1100:5.2f} avg {float(r['AverageNs'])/1e3:7.2f} per step {per:7.2f}")
print('  sum', tot)
PY
  rm -f gpurun_out/prof_small_$c/trace_kernel_trace.csv
done
