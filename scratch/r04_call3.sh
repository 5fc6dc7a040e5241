#!/bin/bash
# r04 call 3: id lists after the padded counters / atomics-first / stores-behind-the-barrier rework: A/B + stamps
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c3; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_range_id_lists" > $O/pytest_lists.txt 2>&1 || { tail -30 $O/pytest_lists.txt; exit 1; }
tail -2 $O/pytest_lists.txt
timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py -x -q -m gpu -k "composite or fused_launches" > $O/pytest_trainer.txt 2>&1 || { tail -30 $O/pytest_trainer.txt; exit 1; }
tail -2 $O/pytest_trainer.txt
for i in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_lists_$i.json 2> $O/bench_lists_$i.err
  TT_ID_BUCKETS=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_scan_$i.json 2> $O/bench_scan_$i.err
done
TT_LIB_PATH=$R/scratch/variants/stamps.so timeout -k 10 200 python scratch/opt_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/stamps_lists.txt
TT_ID_BUCKETS=0 TT_LIB_PATH=$R/scratch/variants/stamps.so timeout -k 10 200 python scratch/opt_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/stamps_scan.txt
echo done
