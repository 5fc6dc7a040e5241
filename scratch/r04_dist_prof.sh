#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_distprof; rm -rf $out; mkdir -p $out
export TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/stdout.txt 2>$out/stderr.txt
python3 - <<'PY'
import csv,re,glob
f=glob.glob("gpurun_out/r04_distprof/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:28]:
    n=r["Name"].replace("void ","").replace("(anonymous namespace)::","")
    m=re.search(r"(\w+_kernel\w*<[^>]*>|\w+_kernel\w*|nccl\w+|\w+)", n)
    print(f'{(m.group(1) if m else n)[:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
