#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/gpu_tests5.log 2>&1; rc=$?; echo "rc=$rc" >> $out/gpu_tests5.log
tail -15 $out/gpu_tests5.log
[ $rc -ne 0 ] && exit 1
bash scratch/r03_ab_opt.sh head main head main
