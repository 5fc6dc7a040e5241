#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "id_encoder or batch_metrics" 2>&1 | tail -15
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $out/gpu_tests21.log 2>&1; rc=$?; echo "rc=$rc" >> $out/gpu_tests21.log
tail -4 $out/gpu_tests21.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/gpu_tests21.log | head -30; exit 1; }
python bench.py --steps 100 --warmup 20 > $out/bench_try.json 2> $out/bench_try.err; tail -c 1500 $out/bench_try.json; tail -3 $out/bench_try.err
