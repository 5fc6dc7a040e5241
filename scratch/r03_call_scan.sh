#!/bin/bash
set -e
mkdir -p gpurun_out/r03scan
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -x -q -m gpu -k "optimizer or train or step or sharded" > gpurun_out/r03scan/tests.log 2>&1 || { tail -40 gpurun_out/r03scan/tests.log; exit 1; }
tail -3 gpurun_out/r03scan/tests.log
bash scratch/r03_ab_opt.sh nopf main nopf main 2>&1 | grep -v "latest WGs\|dense blocks" | tail -30
