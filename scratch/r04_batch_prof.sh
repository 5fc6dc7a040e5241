#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for B in ${BATCHES:-8192 8200}; do
  out=gpurun_out/r04_batchprof_$B; rm -rf $out; mkdir -p $out
  B=$B rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 scratch/r04_batch_prof.py > $out/stdout.txt 2>$out/stderr.txt
  echo "== batch $B"
  python3 - $B <<'PY'
import csv,re,glob,sys
f=glob.glob(f"gpurun_out/r04_batchprof_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:12]:
    n=r["Name"].replace("void ","").replace("(anonymous namespace)::","")
    m=re.search(r"(\w+_kernel\w*<[^>]*>|\w+_kernel\w*|\w+)", n)
    print(f'  {(m.group(1) if m else n)[:56]:56s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
done
