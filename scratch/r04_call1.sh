#!/bin/bash
# r04 call 1: (a) does hipExtLaunchKernelGGL's event pair report the dispatch duration rocprofv3 reports?  (b) slab shapes
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c1; mkdir -p $O
cd $R
python scratch/r04_slab.py 8192x8192x128 2048x16384x128 4096x32768x256 > $O/slab_ext.jsonl 2> $O/slab_ext.err
TT_PROF_BRACKETS=1 python scratch/r04_slab.py 8192x8192x128 2048x16384x128 > $O/slab_brackets.jsonl 2> $O/slab_brackets.err
bash scratch/prof_any.sh r04c1_slab scratch/r04_slab.py 8192x8192x128 2048x16384x128 4096x32768x256 > $O/slab_rocprof_kstats.txt 2>&1
cp $R/gpurun_out/prof_r04c1_slab/stdout.txt $O/slab_under_rocprof.jsonl
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
TT_BENCH_TAGS=score_fused,score_bwd,optimizer,dense_fwd,dense_bwd python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_alltags.json 2> $O/bench_alltags.err
echo done
