#!/bin/bash
# same-box A/B of whole-library variants under rocprof: scratch/r03_ab_lib.sh <kernel substrings, comma separated> variant...   (main = the in-tree library)
keys=$(echo $1 | tr ',' ' '); shift
k=0
for v in "$@"; do
  k=$((k+1))
  if [ "$v" = main ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$PWD/scratch/variants/$v.so; fi
  bash scratch/prof.sh abl${k}_$v --steps 200 --warmup 20 > /dev/null 2>&1
  echo "== $k $v"; python scratch/kstats.py gpurun_out/prof_abl${k}_$v/trace_kernel_stats.csv $keys
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_abl${k}_$v/bench.json
  rm -f gpurun_out/prof_abl${k}_$v/trace_kernel_trace.csv
done
