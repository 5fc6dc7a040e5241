#!/bin/bash
# r04 call 9: dW tiles fed straight from global memory (no LDS operand tiles): parity, stamps, A/B against the LDS-tile form
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c9; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense or lookup_fused or relu_sign or riding or fused_tower" > $O/pytest_parity.txt 2>&1 || { tail -40 $O/pytest_parity.txt; exit 1; }
tail -2 $O/pytest_parity.txt
timeout -k 10 900 python -m pytest tests/test_gpu_trainer.py -x -q -m gpu -k "match_oracle or composite or fused_launches or world1 or category or asymmetric" > $O/pytest_trainer.txt 2>&1 || { tail -40 $O/pytest_trainer.txt; exit 1; }
tail -2 $O/pytest_trainer.txt
TT_LIB_PATH=$R/scratch/variants/gstamps.so timeout -k 10 300 python scratch/gemm_stamps_bwd.py 2>&1 | grep -v amdgpu.ids > $O/gemm_stamps_bwd.txt
cat $O/gemm_stamps_bwd.txt
for i in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_direct_$i.json 2> $O/bench_direct_$i.err
  TT_LIB_PATH=$R/scratch/variants/dwtile.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_tile_$i.json 2> $O/bench_tile_$i.err
done
for f in bench_direct_1 bench_tile_1 bench_direct_2 bench_tile_2; do python - <<PY
import json
d=json.load(open('$O/$f.json')); g=d['roofline_gemm']
print('$f', 'ms/step', round(d['ms_per_step'],4), 'towers us', round(g['us_per_step'],2), 'frac', round(g['frac'],3), 'loss', d['loss_per_pair'])
PY
done
