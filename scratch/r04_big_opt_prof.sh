#!/bin/bash
# kernel durations of the long-list plan / optimizer forms under rocprofv3 (the event-bracket numbers of r04_big_opt_time.py include the host's call rate)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_bigprof; rm -rf $out; mkdir -p $out
TT_BIG_CASES=${TT_BIG_CASES:-0,1} rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 scratch/r04_big_opt_time.py > $out/stdout.txt 2>$out/stderr.txt
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("part_sort", "lds_sort", "merge_rank", "optimizer", "sparse_apply")):
        short = n.split("(")[0][-70:]
        print(f'{short:72s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
