#!/bin/bash
# r03 call 2: optimizer launch with rows requested before the ranking: tests, stamps, rocprof of the cfg3 step
set -x
out=gpurun_out/r03; mkdir -p $out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/gpu_tests2.log 2>&1; rc=$?; echo "rc=$rc" >> $out/gpu_tests2.log
[ $rc -ne 0 ] && { tail -30 $out/gpu_tests2.log; exit 1; }
TT_LIB_PATH=$PWD/scratch/variants/stamps.so timeout -k 10 200 python scratch/opt_stamps.py > $out/opt_stamps.txt 2>&1
timeout -k 10 300 python bench.py > $out/bench_cfg3_a.json 2> $out/bench_cfg3_a.err
bash scratch/prof.sh r03a --steps 200 --warmup 20 > /dev/null 2>&1
cp gpurun_out/prof_r03a/trace_kernel_stats.csv $out/bench_cfg3_kernel_stats_a.csv
python scratch/timeline.py gpurun_out/prof_r03a/trace_kernel_trace.csv > $out/bench_cfg3_timeline_a.txt 2>&1
rm -f gpurun_out/prof_r03a/trace_kernel_trace.csv
cat $out/opt_stamps.txt; tail -3 $out/gpu_tests2.log; cat $out/bench_cfg3_timeline_a.txt
