#!/bin/bash
# usage: scratch/prof_any.sh <tag> <script.py> [args...]   -> gpurun_out/prof_<tag>/ (kernel trace + stats of a python script)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 $GRAFT_REPO_ROOT/"$@" > $out/stdout.txt 2> $out/stderr.txt
python3 $GRAFT_REPO_ROOT/scratch/kstats.py $out/trace_kernel_stats.csv 12
