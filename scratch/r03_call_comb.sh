#!/bin/bash
set -e
mkdir -p gpurun_out/r03comb
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -x -q -m gpu -k "retrieval or rescale or hard or train or step" > gpurun_out/r03comb/tests.log 2>&1 || { tail -40 gpurun_out/r03comb/tests.log; exit 1; }
tail -3 gpurun_out/r03comb/tests.log
bash scratch/r03_call_small.sh 2>&1 | grep "ms_per_step\|fused_combine\|sum"
bash scratch/prof.sh comb3 --steps 200 --warmup 20 > /dev/null 2>&1
python scratch/kstats.py gpurun_out/prof_comb3/trace_kernel_stats.csv fused_combine reduce_slabs
python -c "
import json; print('cfg3 ms_per_step', json.loads(open('gpurun_out/prof_comb3/bench.json').read().strip().splitlines()[-1])['ms_per_step'])"
rm -f gpurun_out/prof_comb3/trace_kernel_trace.csv
