#!/bin/bash
for v in main slabrows64; do
  if [ "$v" = main ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$PWD/scratch/variants/$v.so; fi
  for c in cfg1 cfg2 cfg4; do
    st=2000; [ $c = cfg4 ] && st=50
    echo -n "$v $c "; timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --steps $st --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], (d.get('roofline_gemm') or {}).get('us_per_step'))"
  done
done
