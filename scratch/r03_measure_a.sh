#!/bin/bash
# Round-3 measurement set, part A -> gpurun_out/r03m/ (copied into profiles/r03_* afterwards): bench lines of every config,
# the one-rank sharded legs, rocprof kernel trace + timeline, PMC passes
out=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "== bench cfg3 (default flags)"; timeout -k 10 300 python bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err || echo FAILED
for c in cfg1 cfg2; do
  echo "== bench $c"; timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --steps 2000 --warmup 200 > $out/bench_$c.json 2>/dev/null || echo FAILED
  echo "== bench $c --graph"; timeout -k 10 200 python bench.py --config $c --graph --no-cpu-baseline --steps 2000 --warmup 200 > $out/bench_${c}_graph.json 2>/dev/null || echo FAILED
done
echo "== bench cfg3 --graph"; timeout -k 10 200 python bench.py --graph --no-cpu-baseline > $out/bench_cfg3_graph.json 2>/dev/null || echo FAILED
echo "== bench cfg4 (un-sharded, one GPU)"; timeout -k 10 300 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_cfg4.json 2>/dev/null || echo FAILED
echo "== bench cfg5 (un-sharded, one GPU)"; timeout -k 10 400 python bench.py --config cfg5 --steps 20 --warmup 5 --ids Z --no-cpu-baseline > $out/bench_cfg5.json 2>/dev/null || echo FAILED
echo "== sharded trainer, one rank: no collectives (cfg3), forced RCCL calls (cfg3/4/5)"
TT_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --config cfg3 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_cfg3_nocoll.json || echo FAILED
for c in cfg3 cfg4 cfg5; do
  TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config $c --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_$c.json || echo FAILED
done
echo "== rocprof kernel trace"
bash scratch/prof.sh r03 --steps 200 --warmup 20 > /dev/null 2>&1
cp gpurun_out/prof_r03/trace_kernel_stats.csv $out/bench_cfg3_kernel_stats.csv; cp gpurun_out/prof_r03/bench.json $out/bench_cfg3_under_rocprof.json
python scratch/timeline.py gpurun_out/prof_r03/trace_kernel_trace.csv > $out/bench_cfg3_timeline.txt 2>&1
rm -f gpurun_out/prof_r03/trace_kernel_trace.csv
echo "== PMC passes"
bash scratch/prof_pmc.sh r03 --steps 40 --warmup 10 > /dev/null 2>&1
python scratch/pmc_summary.py gpurun_out/pmc_r03 $out/pmc_cfg3_sgd.json
rm -rf gpurun_out/pmc_r03/*/pmc_counter_collection.csv gpurun_out/pmc_r03/*/*kernel_trace.csv
ls -la $out
for f in bench_cfg1 bench_cfg2 bench_cfg3 bench_cfg4 bench_cfg5 dist1_cfg3_nocoll dist1_cfg3 dist1_cfg4 dist1_cfg5; do python - <<PY
import json
try:
    d = json.loads(open('$out/$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['ms_per_step'], 5), round(d['value']), (d.get('roofline_alt') or {}).get('ms_per_step_alt'))
except Exception as e:
    print('$f', 'unreadable', e)
PY
done
tail -14 $out/bench_cfg3_timeline.txt
# (appended for the re-run after the last kernel change of the round: the per-workgroup stamps and the GEMM launch microbench)
if [ -f scratch/variants/stamps.so ]; then
  TT_LIB_PATH=$PWD/scratch/variants/stamps.so timeout -k 10 200 python scratch/opt_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-600 > $out/optimizer_stamps.txt
  TT_LIB_PATH=$PWD/scratch/variants/stamps.so timeout -k 10 200 python scratch/tower_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-700 > $out/tower_stamps.txt
fi
timeout -k 10 100 python scratch/bench_gemm.py > $out/gemm_launches.json 2>/dev/null
head -2 $out/tower_stamps.txt
