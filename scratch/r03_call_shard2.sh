#!/bin/bash
set -e
mkdir -p gpurun_out/r03shard
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py tests/test_gpu_sharded_world2.py -x -q -m gpu -k "rout or sharded or optimizer or sort or plan or out_of_range or distributed" > gpurun_out/r03shard/tests.log 2>&1 || { tail -40 gpurun_out/r03shard/tests.log; exit 1; }
tail -3 gpurun_out/r03shard/tests.log
bash scratch/r03_call_shard.sh 2>&1 | grep -v "amdgpu.ids\|socket.cpp"
