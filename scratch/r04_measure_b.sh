#!/bin/bash
# Round-4 measurement set, part B: the lines that the release-ticket build of part A had slowed (cfg5: plan + large-list apply), cfg1 again
out=$GRAFT_REPO_ROOT/gpurun_out/r04m; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --config cfg5 --steps 20 --warmup 5 --ids Z --no-cpu-baseline > $out/bench_cfg5.json 2>/dev/null || echo FAILED
TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config cfg5 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_cfg5.json || echo FAILED
for pf in 1 2; do
  TT_PREFETCH=$pf TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config cfg5 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_cfg5_prefetch$pf.json || echo FAILED
done
TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config cfg5 --negatives local --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_cfg5_local.json || echo FAILED
for i in 1 2 3; do timeout -k 10 200 python bench.py --config cfg1 --no-cpu-baseline --steps 2000 --warmup 200 > $out/bench_cfg1_$i.json 2>/dev/null; done
cp $out/bench_cfg1_2.json $out/bench_cfg1.json
timeout -k 10 200 python bench.py --config cfg1 --graph --no-cpu-baseline --steps 2000 --warmup 200 > $out/bench_cfg1_graph.json 2>/dev/null || echo FAILED
timeout -k 10 200 python bench.py --config cfg2 --graph --no-cpu-baseline --steps 2000 --warmup 200 > $out/bench_cfg2_graph.json 2>/dev/null || echo FAILED
timeout -k 10 200 python bench.py --graph --no-cpu-baseline > $out/bench_cfg3_graph.json 2>/dev/null || echo FAILED
for f in bench_cfg5 dist1_cfg5 dist1_cfg5_prefetch1 dist1_cfg5_prefetch2 dist1_cfg5_local bench_cfg1_1 bench_cfg1_2 bench_cfg1_3 bench_cfg1_graph bench_cfg2_graph bench_cfg3_graph; do python - <<PY
import json
try:
    d = json.loads(open('$out/$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['ms_per_step'], 5), round(d['value']), (d.get('roofline_alt') or {}).get('ms_per_step_alt'), round(d['roofline_hbm']['optimizer_launch_us'],2) if 'roofline_hbm' in d else '')
except Exception as e:
    print('$f', 'unreadable', e)
PY
done
