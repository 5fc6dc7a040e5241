#!/bin/bash
# dW slabs 32 (HEAD) vs 16: same box, alternating, cfg3 default run
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in head s16; do
    if [ $v = head ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$GRAFT_REPO_ROOT/scratch/variants/$v.so; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/slabs_$v.json
    python - <<PY
import json
d=json.loads(open("gpurun_out/slabs_$v.json").read()); h=d["roofline_hbm"]; g=d.get("roofline_gemm") or {}
print("$v", round(d["ms_per_step"],5), h["optimizer_launch_us"], g.get("launch_us"), g.get("frac"))
PY
  done
done
