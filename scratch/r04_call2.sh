#!/bin/bash
# r04 call 2: row-range id lists (forward lookup -> optimizer launch): parity tests, then the A/B on the cfg3 bench; slab shapes warm
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c2; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_range_id_lists or raw_ids or heavy_hitters" > $O/pytest_lists.txt 2>&1 || { tail -30 $O/pytest_lists.txt; exit 1; }
tail -3 $O/pytest_lists.txt
timeout -k 10 900 python -m pytest tests/test_gpu_trainer.py -x -q -m gpu -k "composite or fused_launches or match_oracle or world1 or cfg3" > $O/pytest_trainer.txt 2>&1 || { tail -30 $O/pytest_trainer.txt; exit 1; }
tail -3 $O/pytest_trainer.txt
for i in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_lists_$i.json 2> $O/bench_lists_$i.err
  TT_ID_BUCKETS=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_scan_$i.json 2> $O/bench_scan_$i.err
done
python scratch/r04_slab.py 8192x8192x128 2048x16384x128 4096x32768x256 > $O/slab_warm.jsonl 2> $O/slab_warm.err
echo done
