#!/bin/bash
# bf16x3 at dim 256 (wave-pair split) + FWD/RANK in bf16x3: tests, then the scorer microbench with both pass-2 forms
set -e
mkdir -p gpurun_out/r03bx3
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "retrieval or rescale or hard" > gpurun_out/r03bx3/tests.log 2>&1 || { tail -40 gpurun_out/r03bx3/tests.log; exit 1; }
tail -3 gpurun_out/r03bx3/tests.log
timeout -k 10 300 python scratch/bench_score.py 8192x128 8192x256 32768x256 > gpurun_out/r03bx3/score_recompute.jsonl 2> gpurun_out/r03bx3/score_recompute.err
TT_BX3_KEEP256=1 timeout -k 10 300 python scratch/bench_score.py 8192x256 32768x256 > gpurun_out/r03bx3/score_keep.jsonl 2> gpurun_out/r03bx3/score_keep.err
cat gpurun_out/r03bx3/score_recompute.jsonl gpurun_out/r03bx3/score_keep.jsonl
