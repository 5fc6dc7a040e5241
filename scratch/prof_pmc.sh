#!/bin/bash
# usage: scratch/prof_pmc.sh <tag> [bench args]  -> gpurun_out/pmc_<tag>/{fetch,write,mfma}/...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
base=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  out=$base/$name; mkdir -p $out
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err || { echo "pass $name failed"; tail -5 $out/bench.err; }
  ls $out | head
done
