#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 scratch/r04_bwd2_check.py || exit 1
DROP=0.1 timeout -k 10 300 python3 scratch/r04_bwd2_check.py || exit 1
out=gpurun_out/r04_bwd2prof; rm -rf $out; mkdir -p $out
STEPS=60 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 scratch/r04_bwd2_check.py > $out/stdout.txt 2>$out/stderr.txt
python3 - <<'PY'
import csv,re,glob
f=glob.glob("gpurun_out/r04_bwd2prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "bwd" in n or "fwd2" in n:
        m=re.search(r"(\w+_kernel\w*<[^>]*>|\w+_kernel\w*)", n)
        print(f'{m.group(1) if m else n[:60]:40s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
