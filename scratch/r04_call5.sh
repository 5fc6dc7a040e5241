#!/bin/bash
# r04 call 5: dense-update riders in the backward launches (+ id lists): parity, A/B, stamps
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c5; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_trainer.py -x -q -m gpu -k "composite or fused_launches or match_oracle or world1 or cfg3 or category" > $O/pytest_trainer.txt 2>&1 || { tail -30 $O/pytest_trainer.txt; exit 1; }
tail -2 $O/pytest_trainer.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_range_id_lists or dense" > $O/pytest_parity.txt 2>&1 || { tail -30 $O/pytest_parity.txt; exit 1; }
tail -2 $O/pytest_parity.txt
for i in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_riders_$i.json 2> $O/bench_riders_$i.err
  TT_DENSE_RIDERS=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_noriders_$i.json 2> $O/bench_noriders_$i.err
  TT_DENSE_RIDERS=0 TT_ID_BUCKETS=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_r03form_$i.json 2> $O/bench_r03form_$i.err
done
TT_LIB_PATH=$R/scratch/variants/stamps.so timeout -k 10 200 python scratch/opt_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 > $O/stamps_riders.txt
echo done
