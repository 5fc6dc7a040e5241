#!/bin/bash
# usage: scratch/prof.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
ls -R $out | head -30
