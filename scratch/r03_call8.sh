#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/gpu_tests8.log 2>&1; rc=$?; echo "rc=$rc" >> $out/gpu_tests8.log
tail -5 $out/gpu_tests8.log
[ $rc -ne 0 ] && exit 1
bash scratch/r03_ab_opt.sh head main head main > /dev/null 2>&1
for f in $out/ab_opt_[0-9]*; do echo "== $f"; cat $f; done
for kv in 1 0; do
  export HIP_FORCE_DEV_KERNARG=$kv
  bash scratch/prof.sh kernarg$kv --steps 200 --warmup 20 > /dev/null 2>&1
  echo "== HIP_FORCE_DEV_KERNARG=$kv"
  python scratch/timeline.py gpurun_out/prof_kernarg$kv/trace_kernel_trace.csv 2>&1 | tail -12
  rm -f gpurun_out/prof_kernarg$kv/trace_kernel_trace.csv
  for i in 1 2; do python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; done
done
unset HIP_FORCE_DEV_KERNARG
timeout -k 10 300 python bench_kernels.py --only table > $out/kernels_nt.jsonl 2>/dev/null; cut -c1-160 $out/kernels_nt.jsonl
