#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c11; mkdir -p $O
cd $R
for i in 1 2; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_head_$i.json 2> $O/bench_head_$i.err
  TT_GEMM_DX_PAIR=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_nopair_$i.json 2> $O/bench_nopair_$i.err
done
for f in bench_head_1 bench_nopair_1 bench_head_2 bench_nopair_2; do python - <<PY
import json
d=json.load(open('$O/$f.json')); g=d['roofline_gemm']
print('$f', 'ms/step', round(d['ms_per_step'],4), 'towers us', round(g['us_per_step'],2), 'frac', round(g['frac'],3))
PY
done
