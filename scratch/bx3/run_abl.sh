#!/bin/bash
for v in base 2 4 6 15; do
  if [ $v = base ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$GRAFT_REPO_ROOT/scratch/bx3/libabl$v.so; fi
  python scratch/bench_score.py 2>/dev/null | head -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('abl $v', 'fused', round(d['bf16x3']['fused_us'],1), 'bwd', round(d['bf16x3']['bwd_us'],1))"
done
