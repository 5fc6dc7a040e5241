#!/bin/bash
set -e
mkdir -p gpurun_out/r03qt
TT_LIB_PATH=$PWD/scratch/variants/qt.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "retrieval or rescale" > gpurun_out/r03qt/tests.log 2>&1 || { tail -30 gpurun_out/r03qt/tests.log; exit 1; }
tail -2 gpurun_out/r03qt/tests.log
for v in main qt main qt; do
  if [ "$v" = main ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$PWD/scratch/variants/$v.so; fi
  bash scratch/prof.sh ab_$v --steps 200 --warmup 20 > /dev/null 2>&1
  echo "-- $v"
  python scratch/kstats.py gpurun_out/prof_ab_$v/trace_kernel_stats.csv "score_kernel<128, 5" "score_kernel<128, 4"
  python -c "
import json; print('ms_per_step', json.loads(open('gpurun_out/prof_ab_$v/bench.json').read().strip().splitlines()[-1])['ms_per_step'])"
  rm -f gpurun_out/prof_ab_$v/trace_kernel_trace.csv
done
