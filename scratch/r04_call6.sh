#!/bin/bash
# r04 call 6: wave-0 list emission; riders entry test; --config ref / --engine ops lines
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c6; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_range_id_lists or riding or fused_tower or lookup_fused" > $O/pytest_parity.txt 2>&1 || { tail -30 $O/pytest_parity.txt; exit 1; }
tail -2 $O/pytest_parity.txt
timeout -k 10 900 python -m pytest tests/test_gpu_trainer.py -x -q -m gpu -k "composite or fused_launches or facade or custom_ops or retrieval_task" > $O/pytest_trainer.txt 2>&1 || { tail -30 $O/pytest_trainer.txt; exit 1; }
tail -2 $O/pytest_trainer.txt
for i in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_lists_$i.json 2> $O/bench_lists_$i.err
  TT_ID_BUCKETS=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_scan_$i.json 2> $O/bench_scan_$i.err
done
python bench.py --config ref --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_ref.json 2> $O/bench_ref.err
python bench.py --config ref --engine ops --steps 200 --warmup 20 > $O/bench_ref_ops.json 2> $O/bench_ref_ops.err
python bench.py --config cfg3 --engine ops --steps 100 --warmup 10 > $O/bench_cfg3_ops.json 2> $O/bench_cfg3_ops.err
echo done
