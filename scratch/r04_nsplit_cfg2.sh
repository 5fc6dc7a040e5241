#!/bin/bash
cd $GRAFT_REPO_ROOT
for q in 16 8 32 24 12; do for cs in 16 8 32; do
  TT_NSPLIT_Q=$q TT_NSPLIT_CS=$cs timeout -k 10 200 python bench.py --config cfg2 --no-cpu-baseline --steps 1500 --warmup 150 2>/dev/null | grep '^{' > gpurun_out/ns_cfg2.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/ns_cfg2.json").read()); r=d["roofline"]
print("ns_q $q ns_cs $cs", round(d["ms_per_step"],5), round(r.get("avg_launch_us",0),1), round(r.get("other_pass_avg_launch_us",0) or 0,1))
PY
done; done
