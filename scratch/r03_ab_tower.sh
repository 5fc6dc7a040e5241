#!/bin/bash
# tests of the sign-bit / tower epilogue, then same-box A/B of the tower launches (variant libs as arguments, "main" = in-tree)
set -e
mkdir -p gpurun_out/r03tw
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -x -q -m gpu -k "tower or sign_bits or relu or dense or train or step" > gpurun_out/r03tw/tests.log 2>&1 || { tail -30 gpurun_out/r03tw/tests.log; exit 1; }
tail -2 gpurun_out/r03tw/tests.log
for v in "$@"; do
  if [ "$v" = main ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$PWD/scratch/variants/$v.so; fi
  bash scratch/prof.sh ab_$v --steps 200 --warmup 20 > /dev/null 2>&1
  echo "-- $v"
  python scratch/kstats.py gpurun_out/prof_ab_$v/trace_kernel_stats.csv tower_fwd2 gemm_bwd
  python -c "
import json; print('ms_per_step', json.loads(open('gpurun_out/prof_ab_$v/bench.json').read().strip().splitlines()[-1])['ms_per_step'])"
  rm -f gpurun_out/prof_ab_$v/trace_kernel_trace.csv
done
