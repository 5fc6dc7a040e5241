#!/bin/bash
# A/B of the scorer epilogue (packed v_pk_* vs scalar f32 VALU) on ONE box, alternating runs: prints the live hipEvent
# averages of both score kernels and the step time per run.
for i in 1 2 3; do
  for v in pk scalar; do
    if [ $v = scalar ]; then export TT_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libtwotower_scalar_epi.so; else unset TT_LIB_PATH; fi
    python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['ms_per_step'],4), 'fused_us', round(r['avg_launch_us'],2), 'bwd_us', round(r['other_pass']['avg_launch_us'],2))"
  done
done
