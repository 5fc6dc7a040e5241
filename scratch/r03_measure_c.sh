#!/bin/bash
# part C: full GPU suite at HEAD, then the bench lines whose second (bf16x3) leg or work_model changed
out=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1 || { tail -40 $out/gpu_tests.log; exit 1; }
tail -3 $out/gpu_tests.log
echo "== bench cfg3 (default flags)"; timeout -k 10 300 python bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err || echo FAILED
echo "== bench cfg4"; timeout -k 10 300 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_cfg4.json 2>/dev/null || echo FAILED
TT_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --config cfg3 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_cfg3_nocoll.json || echo FAILED
for c in cfg3 cfg4 cfg5; do
  TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config $c --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_$c.json || echo FAILED
done
for f in bench_cfg3 bench_cfg4 dist1_cfg3_nocoll dist1_cfg3 dist1_cfg4 dist1_cfg5; do python - <<PY
import json
try:
    d = json.loads(open('$out/$f.json').read().strip().splitlines()[-1])
    ra = d.get('roofline_alt') or {}
    wm = d.get('work_model') or {}
    print('$f', round(d['ms_per_step'], 5), round(d['value']), 'alt', ra.get('ms_per_step_alt'), ra.get('value_alt'), ra.get('avg_launch_us'), 'wm', wm.get('ms_expected_from_n1'), wm.get('scorer_flops_per_gpu_vs_n1'),
          'gemm', (d.get('roofline_gemm') or {}).get('rocprof_frac'), 'hbm', (d.get('roofline_hbm') or {}).get('rocprof_optimizer_launch_us'))
except Exception as e:
    print('$f', 'unreadable', e)
PY
done
