#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-head noacq norel nosync}; do
  if [ $v = head ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$GRAFT_REPO_ROOT/scratch/variants/$v.so; fi
  out=gpurun_out/r04_bwd2var_$v; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 scratch/r04_bwd2_time.py > $out/stdout.txt 2>$out/stderr.txt
  python3 - $v <<'PY'
import csv,re,glob,sys
f=glob.glob(f"gpurun_out/r04_bwd2var_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "bwd2" in n:
        print(sys.argv[1], f'calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
done
