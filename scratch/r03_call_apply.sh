#!/bin/bash
set -e
mkdir -p gpurun_out/r03apply
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse or optimizer or plan" > gpurun_out/r03apply/tests.log 2>&1 || { tail -40 gpurun_out/r03apply/tests.log; exit 1; }
tail -2 gpurun_out/r03apply/tests.log
for v in oldscan main applyw6; do
  if [ "$v" = main ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$PWD/scratch/variants/$v.so; fi
  echo "== $v"
  timeout -k 10 300 python bench_kernels.py --only table 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if 'sparse_apply' in d['kernel']: print(' ', d['kernel'], d['ids'], round(d['us'],1), 'us', round(d['frac_hbm_8TBs'],3))"
done
