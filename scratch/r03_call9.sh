#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/gpu_tests9.log 2>&1; rc=$?; echo "rc=$rc" >> $out/gpu_tests9.log
tail -5 $out/gpu_tests9.log
[ $rc -ne 0 ] && { grep -n "Error\|error" $out/gpu_tests9.log | head -20; exit 1; }
for c in cfg1 cfg2 cfg3; do
  for comp in 1 0; do
    echo "== $c composite=$comp"
    TT_COMPOSITE_STEP=$comp python bench.py --config $c --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.]*'
  done
done
python scratch/host_time.py 2>&1 | tail -8
