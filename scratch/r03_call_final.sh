#!/bin/bash
set -e
mkdir -p gpurun_out/r03m
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03m/gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03m/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r03m/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()"
