#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_opsprof; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 bench.py --engine ops --steps 100 --warmup 10 --no-cpu-baseline > $out/stdout.txt 2>$out/stderr.txt
tail -c 400 $out/stdout.txt; echo
python3 - <<'PY'
import csv,re,glob
f=glob.glob("gpurun_out/r04_opsprof/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:32]:
    n=r["Name"].replace("void ","").replace("(anonymous namespace)::","")
    m=re.search(r"(\w+_kernel\w*<[^>]*>|\w+_kernel\w*|[\w:]+)", n)
    per=float(r["TotalDurationNs"])/110/1e3
    print(f'{(m.group(1) if m else n)[:70]:70s} calls/step {int(r["Calls"])/110:5.1f} avg {float(r["AverageNs"])/1e3:8.1f} us  per step {per:7.1f}')
PY
