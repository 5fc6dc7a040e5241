#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c15; mkdir -p $O
cd $R
TT_LIB_PATH=$R/scratch/variants/dwtile.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_tile32.json 2>/dev/null
TT_LIB_PATH=$R/scratch/variants/dwtile16.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_tile16.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_direct.json 2>/dev/null
python - <<PY
import json
for n in ('tile32','tile16','direct'):
    print(n, json.load(open('$O/bench_'+n+'.json'))['loss_per_pair'])
PY
