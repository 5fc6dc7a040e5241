#!/bin/bash
# full GPU suite, then cfg5 / cfg3 bench lines + the large-list kernels
set -e
mkdir -p gpurun_out/r03full
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03full/tests.log 2>&1 || { tail -40 gpurun_out/r03full/tests.log; exit 1; }
tail -3 gpurun_out/r03full/tests.log
timeout -k 10 300 python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/r03full/bench_cfg5.json 2> gpurun_out/r03full/bench_cfg5.err
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03full/bench_cfg3.json 2> gpurun_out/r03full/bench_cfg3.err
python - <<'PY'
import json
for n in ("cfg5", "cfg3"):
    d = json.loads(open(f"gpurun_out/r03full/bench_{n}.json").read().strip().splitlines()[-1])
    ra = d.get("roofline_alt") or {}
    h = d["roofline_hbm"]
    print(n, d["value"], d["ms_per_step"], ra.get("value_alt"), ra.get("ms_per_step_alt"), "gemm", (d.get("roofline_gemm") or {}).get("frac"), "hbm", h["frac"], {k: h[k] for k in h if k.endswith("_us")})
PY
timeout -k 10 300 python bench_kernels.py --only table 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print(' ', d['kernel'][:40], d.get('ids'), round(d['us'],1), 'us', round(d.get('frac_hbm_8TBs',0),3))"
