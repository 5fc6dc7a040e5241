#!/bin/bash
# r04 call 4: launch floor probe; id lists with the entries stored at the end of the forward kernel, atomics behind / in front of the row loads
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c4; mkdir -p $O
cd $R
timeout -k 10 300 scratch/variants/launch_floor > $O/floor_warm.txt 2>&1
FLUSH=1 timeout -k 10 300 scratch/variants/launch_floor > $O/floor_flushed.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_range_id_lists" > $O/pytest_lists.txt 2>&1 || { tail -30 $O/pytest_lists.txt; exit 1; }
tail -2 $O/pytest_lists.txt
for i in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_lists_$i.json 2> $O/bench_lists_$i.err
  TT_LIB_PATH=$R/scratch/variants/atomics_first.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_af_$i.json 2> $O/bench_af_$i.err
  TT_ID_BUCKETS=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_scan_$i.json 2> $O/bench_scan_$i.err
done
echo done
