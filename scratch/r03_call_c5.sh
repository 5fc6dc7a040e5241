#!/bin/bash
for f in 1 0; do
  echo -n "cfg5 TT_FUSE_LOOKUP=$f "; TT_FUSE_LOOKUP=$f timeout -k 10 300 python bench.py --config cfg5 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], (d.get('roofline_gemm') or {}).get('us_per_step'), (d.get('roofline_alt') or {}).get('ms_per_step_alt'))"
done
for f in 1 0; do
  echo -n "cfg4 TT_FUSE_LOOKUP=$f "; TT_FUSE_LOOKUP=$f timeout -k 10 300 python bench.py --config cfg4 --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], (d.get('roofline_gemm') or {}).get('us_per_step'))"
done
