#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
python scratch/dbg_opt_ids.py 2>&1 | grep "^n \|first bad" | cut -c1-200
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/gpu_tests7.log 2>&1; rc=$?; echo "rc=$rc" >> $out/gpu_tests7.log
tail -5 $out/gpu_tests7.log
[ $rc -ne 0 ] && exit 1
bash scratch/r03_ab_opt.sh head main head main
