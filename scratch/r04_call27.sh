#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c27; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "row_range_id_lists or lookup_fused or dense" 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests/test_gpu_trainer.py -q -m gpu 2>&1 | tail -3
for c in cfg1 cfg2 ref; do
  for i in 1 2; do
  python bench.py --config $c --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c lists', round(d['ms_per_step'],5), round(d['roofline_hbm']['optimizer_launch_us'],2))"
  TT_ID_BUCKETS=0 python bench.py --config $c --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c scan ', round(d['ms_per_step'],5), round(d['roofline_hbm']['optimizer_launch_us'],2))"
  done
done
python scratch/r04_soak.py 1500 adagrad 2>/dev/null | tail -1
