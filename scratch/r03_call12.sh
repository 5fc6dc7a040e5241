#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_tower" 2>&1 | tail -15
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $out/gpu_tests12.log 2>&1; rc=$?; echo "rc=$rc" >> $out/gpu_tests12.log
tail -5 $out/gpu_tests12.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/gpu_tests12.log | head -30; exit 1; }
for f in 1 0 1 0; do
  export TT_FUSED_TOWER=$f
  bash scratch/prof.sh tower$f --steps 200 --warmup 20 > /dev/null 2>&1
  echo "== TT_FUSED_TOWER=$f"
  python scratch/timeline.py gpurun_out/prof_tower$f/trace_kernel_trace.csv 2>&1 | tail -12
  rm -f gpurun_out/prof_tower$f/trace_kernel_trace.csv
done
