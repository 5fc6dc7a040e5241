#!/bin/bash
# sharded trainer on one rank: no collectives vs plain trainer, kernel stats of both
set -e
out=gpurun_out/r03shard; mkdir -p $out
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 | grep '^{' > $out/plain.json
TT_FORCE_DIST=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 | grep '^{' > $out/shard_nocoll.json
TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 | grep '^{' > $out/shard_coll.json
python - <<'PY'
import json
for n in ("plain", "shard_nocoll", "shard_coll"):
    d = json.loads(open(f"gpurun_out/r03shard/{n}.json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"])
PY
cd /tmp && export TMPDIR=/tmp
export TT_FORCE_DIST=1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 --warmup 10 > $GRAFT_REPO_ROOT/$out/prof_bench.json 2> $GRAFT_REPO_ROOT/$out/prof_bench.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r03shard/prof/trace_kernel_stats.csv')))
tot = 0
for r in rows:
    per_step = float(r['TotalDurationNs']) / 1e3 / 110
    tot += per_step
    if per_step > 0.5:
        print(f"{r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:70]:70s} calls/step {int(r['Calls'])/110:5.2f} avg {float(r['AverageNs'])/1e3:8.2f} us  per step {per_step:8.2f}")
print('sum per step', tot)
PY
rm -f $out/prof/trace_kernel_trace.csv
