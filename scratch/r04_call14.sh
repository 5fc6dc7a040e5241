#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c14; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py -q -m gpu -k "fused_launches or composite" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for st in 20 100; do
  python bench.py --steps $st --warmup 5 --no-cpu-baseline > $O/bench_direct_$st.json 2>/dev/null
  TT_LIB_PATH=$R/scratch/variants/dwtile.so python bench.py --steps $st --warmup 5 --no-cpu-baseline > $O/bench_tile_$st.json 2>/dev/null
  python - <<PY
import json
a=json.load(open('$O/bench_direct_$st.json')); b=json.load(open('$O/bench_tile_$st.json'))
print('steps $st: loss/pair direct', a['loss_per_pair'], 'tile', b['loss_per_pair'], 'diff', a['loss_per_pair']-b['loss_per_pair'])
PY
done
