#!/bin/bash
# scorer workgroup-count / wave-count variants at the small configs (and cfg4's slab shape via bench_score)
for v in main wgs256 wgs1024 f8 b4; do
  if [ "$v" = main ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$PWD/scratch/variants/$v.so; fi
  for c in cfg2 cfg1; do
    echo -n "$v $c "; timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --steps 2000 --warmup 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['ms_per_step'],5), 'dominant', round(r['avg_launch_us'],1), 'other', round(r['other_pass']['avg_launch_us'],1))"
  done
done
