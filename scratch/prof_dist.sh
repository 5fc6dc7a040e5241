#!/bin/bash
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
export TT_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
