#!/bin/bash
# usage: scratch/prof_pmc_sq.sh <tag>   -> gpurun_out/pmcsq_<tag>/{lds,issue}/  (stall attribution of the score kernels)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
base=$GRAFT_REPO_ROOT/gpurun_out/pmcsq_$tag
for pass in "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES" "issue:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  out=$base/$name; mkdir -p $out
  rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex score_kernel --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 12 --warmup 3 > $out/bench.json 2> $out/bench.err || { echo "pass $name failed"; tail -5 $out/bench.err; }
  ls $out | head
done
