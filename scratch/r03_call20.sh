#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $out/gpu_tests20.log 2>&1; rc=$?; echo "rc=$rc" >> $out/gpu_tests20.log
tail -4 $out/gpu_tests20.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/gpu_tests20.log | head -30; exit 1; }
bash scratch/r03_ab_lib.sh tower_fwd2,gemm_ main narrow main narrow
TT_FUSED_TOWER=0 bash scratch/r03_ab_lib.sh gemm_kernel main narrow
