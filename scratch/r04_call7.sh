#!/bin/bash
# r04 call 7: slab shapes with 8-wave pass-1 workgroups (variant) vs 4-wave (HEAD); cfg1 / cfg2 kernel breakdown
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c7; mkdir -p $O
cd $R
for i in 1 2; do
  python scratch/r04_slab.py 8192x8192x128 2048x16384x128 4096x4096x64 > $O/slab_head_$i.jsonl 2> $O/slab_head_$i.err
  TT_LIB_PATH=$R/scratch/variants/fuseds8.so python scratch/r04_slab.py 8192x8192x128 2048x16384x128 4096x4096x64 > $O/slab_f8_$i.jsonl 2> $O/slab_f8_$i.err
done
bash scratch/prof.sh r04c7_cfg2 --config cfg2 --steps 200 --warmup 20 > /dev/null 2>&1
python scratch/kstats.py gpurun_out/prof_r04c7_cfg2/trace_kernel_stats.csv kernel > $O/cfg2_kstats.txt
cp gpurun_out/prof_r04c7_cfg2/bench.json $O/cfg2_under_rocprof.json
bash scratch/prof.sh r04c7_cfg1 --config cfg1 --steps 400 --warmup 40 > /dev/null 2>&1
python scratch/kstats.py gpurun_out/prof_r04c7_cfg1/trace_kernel_stats.csv kernel > $O/cfg1_kstats.txt
rm -f gpurun_out/prof_r04c7_cfg*/trace_kernel_trace.csv
python bench.py --config cfg2 --steps 400 --warmup 40 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python bench.py --config cfg1 --steps 400 --warmup 40 --no-cpu-baseline > $O/bench_cfg1.json 2> $O/bench_cfg1.err
echo done
