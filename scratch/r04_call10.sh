#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c10; mkdir -p $O
cd $R
TT_LIB_PATH=$R/scratch/variants/gstamps.so timeout -k 10 300 python scratch/gemm_stamps_bwd.py 2>&1 | grep -v amdgpu.ids > $O/gemm_stamps_bwd.txt
cat $O/gemm_stamps_bwd.txt
for i in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_prio_$i.json 2> $O/bench_prio_$i.err
  TT_LIB_PATH=$R/scratch/variants/pf8.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_pf8_$i.json 2> $O/bench_pf8_$i.err
  TT_LIB_PATH=$R/scratch/variants/dwtile.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_tile_$i.json 2> $O/bench_tile_$i.err
done
for f in bench_prio_1 bench_pf8_1 bench_tile_1 bench_prio_2 bench_pf8_2 bench_tile_2; do python - <<PY
import json
d=json.load(open('$O/$f.json')); g=d['roofline_gemm']
print('$f', 'ms/step', round(d['ms_per_step'],4), 'towers us', round(g['us_per_step'],2), 'frac', round(g['frac'],3), 'loss', d['loss_per_pair'])
PY
done
