#!/bin/bash
# r03 call 1: (a) does the lambda-in-loop form of the scorer's plain tile loop reproduce the r02 wrong rows?  (b) sparse_apply A/B
set -x
mkdir -p gpurun_out/r03
export TT_LIB_PATH=$PWD/scratch/variants/lam.so
for i in 1 2 3; do
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "baseline_configs or deterministic" > gpurun_out/r03/lam_tests_$i.log 2>&1; echo "lam rc=$?" >> gpurun_out/r03/lam_tests_$i.log
done
unset TT_LIB_PATH
timeout -k 10 300 python bench_kernels.py --only table > gpurun_out/r03/kernels_new.jsonl 2> gpurun_out/r03/kernels_new.err
TT_LIB_PATH=$PWD/scratch/variants/oldfence.so timeout -k 10 300 python bench_kernels.py --only table > gpurun_out/r03/kernels_oldfence.jsonl 2> gpurun_out/r03/kernels_oldfence.err
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r03/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03/gpu_tests.log
tail -3 gpurun_out/r03/*.log
