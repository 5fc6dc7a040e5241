#!/bin/bash
# Round-4 measurement set, part C: the lines touched by the round's last changes (id lists from layer 0's own forward launch:
# cfg1 / cfg2 / reference config; fused lookup at small batches: reference config; scratch-free chunk sort: cfg5's plan)
out=$GRAFT_REPO_ROOT/gpurun_out/r04m; mkdir -p $out
cd $GRAFT_REPO_ROOT
for c in cfg1 cfg2; do
  timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --steps 2000 --warmup 200 > $out/bench_$c.json 2>/dev/null || echo FAILED
done
timeout -k 10 200 python bench.py --config ref --steps 400 --warmup 40 --no-cpu-baseline > $out/bench_ref.json 2>/dev/null || echo FAILED
timeout -k 10 200 python bench.py --config ref --engine ops --steps 200 --warmup 20 > $out/bench_ref_ops.json 2>/dev/null || echo FAILED
timeout -k 10 400 python bench.py --config cfg5 --steps 20 --warmup 5 --ids Z --no-cpu-baseline > $out/bench_cfg5.json 2>/dev/null || echo FAILED
TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config cfg5 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_cfg5.json || echo FAILED
timeout -k 10 300 python bench.py > $out/bench_cfg3_recheck.json 2>/dev/null || echo FAILED
for f in bench_cfg1 bench_cfg2 bench_ref bench_ref_ops bench_cfg5 dist1_cfg5 bench_cfg3_recheck; do python - <<PY
import json
try:
    d = json.loads(open('$out/$f.json').read().strip().splitlines()[-1])
    h = d.get('roofline_hbm') or {}
    print('$f', round(d['ms_per_step'], 5), round(d['value']), (d.get('roofline_alt') or {}).get('ms_per_step_alt'), h.get('optimizer_launch_us'), h.get('sparse_plan_us'), h.get('frac'))
except Exception as e:
    print('$f', 'unreadable', e)
PY
done
timeout -k 10 900 python -m pytest tests -q -m gpu > $out/gputests.txt 2>&1; tail -4 $out/gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
