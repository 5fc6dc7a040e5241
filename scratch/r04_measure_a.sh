#!/bin/bash
# Round-4 measurement set, part A -> gpurun_out/r04m/ (copied into profiles/r04_* afterwards): bench lines of every config, the
# extra lines (reference config, ops engine), the one-rank sharded legs, rocprof kernel trace + timeline, PMC passes
out=$GRAFT_REPO_ROOT/gpurun_out/r04m; mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "== bench cfg3 (default flags)"; timeout -k 10 300 python bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err || echo FAILED
for c in cfg1 cfg2; do
  echo "== bench $c"; timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --steps 2000 --warmup 200 > $out/bench_$c.json 2>/dev/null || echo FAILED
done
echo "== bench cfg4 (un-sharded, one GPU)"; timeout -k 10 300 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_cfg4.json 2>/dev/null || echo FAILED
echo "== bench cfg5 (un-sharded, one GPU)"; timeout -k 10 400 python bench.py --config cfg5 --steps 20 --warmup 5 --ids Z --no-cpu-baseline > $out/bench_cfg5.json 2>/dev/null || echo FAILED
echo "== extra lines: the reference's own config, both engines; cfg3 through the ops engine"
timeout -k 10 200 python bench.py --config ref --steps 400 --warmup 40 --no-cpu-baseline > $out/bench_ref.json 2>/dev/null || echo FAILED
timeout -k 10 200 python bench.py --config ref --engine ops --steps 200 --warmup 20 > $out/bench_ref_ops.json 2>/dev/null || echo FAILED
timeout -k 10 200 python bench.py --config cfg3 --engine ops --steps 100 --warmup 10 > $out/bench_cfg3_ops.json 2>/dev/null || echo FAILED
echo "== sharded trainer, one rank: no collectives (cfg3), forced RCCL calls (cfg3/4/5; cfg4/5 also with prefetch 1|2 and local negatives)"
TT_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --config cfg3 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_cfg3_nocoll.json || echo FAILED
for c in cfg3 cfg4 cfg5; do
  TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config $c --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_$c.json || echo FAILED
done
for c in cfg4 cfg5; do
  for pf in 1 2; do
    TT_PREFETCH=$pf TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config $c --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_${c}_prefetch$pf.json || echo FAILED
  done
  TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config $c --negatives local --no-cpu-baseline 2>/dev/null | grep '^{' > $out/dist1_${c}_local.json || echo FAILED
done
echo "== scorer at the per-GPU slab shapes of the 8-GPU configurations"
timeout -k 10 300 python scratch/r04_slab.py 8192x8192x128 2048x16384x128 4096x32768x256 4096x4096x64 --prec both > $out/score_slabs.jsonl 2>/dev/null || echo FAILED
bash scratch/prof_any.sh r04_slab scratch/r04_slab.py 2048x16384x128 > $out/score_slab_cfg4_kstats.txt 2>&1
cp gpurun_out/prof_r04_slab/trace_kernel_stats.csv $out/score_slab_cfg4_kernel_stats.csv
echo "== rocprof kernel trace"
bash scratch/prof.sh r04 --steps 200 --warmup 20 > /dev/null 2>&1
cp gpurun_out/prof_r04/trace_kernel_stats.csv $out/bench_cfg3_kernel_stats.csv; cp gpurun_out/prof_r04/bench.json $out/bench_cfg3_under_rocprof.json
python scratch/timeline.py gpurun_out/prof_r04/trace_kernel_trace.csv > $out/bench_cfg3_timeline.txt 2>&1
rm -f gpurun_out/prof_r04/trace_kernel_trace.csv gpurun_out/prof_r04_slab/trace_kernel_trace.csv
echo "== rocprof A/B of the layer-0 launches: lookup fused (default) vs materialised input (TT_FUSE_LOOKUP=0), alternating"
for i in 1 2; do
  for v in 1 0; do
    TT_FUSE_LOOKUP=$v bash scratch/prof.sh r04_lk${v}_$i --steps 200 --warmup 20 > /dev/null 2>&1
    echo "-- TT_FUSE_LOOKUP=$v run $i" >> $out/lookup_share_ab.txt
    python scratch/kstats.py gpurun_out/prof_r04_lk${v}_$i/trace_kernel_stats.csv tower_fwd2 gemm_bwd_kernel gather_kernel optimizer_ids >> $out/lookup_share_ab.txt
    python -c "
import json; print('ms_per_step', json.loads(open('gpurun_out/prof_r04_lk${v}_$i/bench.json').read().strip().splitlines()[-1])['ms_per_step'])" >> $out/lookup_share_ab.txt
    rm -f gpurun_out/prof_r04_lk${v}_$i/trace_kernel_trace.csv
  done
done
echo "== PMC passes"
bash scratch/prof_pmc.sh r04 --steps 40 --warmup 10 > /dev/null 2>&1
python scratch/pmc_summary.py gpurun_out/pmc_r04 $out/pmc_cfg3_sgd.json
rm -rf gpurun_out/pmc_r04/*/pmc_counter_collection.csv gpurun_out/pmc_r04/*/*kernel_trace.csv
ls -la $out
for f in bench_cfg1 bench_cfg2 bench_cfg3 bench_cfg4 bench_cfg5 bench_ref bench_ref_ops bench_cfg3_ops dist1_cfg3_nocoll dist1_cfg3 dist1_cfg4 dist1_cfg5 dist1_cfg4_prefetch1 dist1_cfg4_prefetch2 dist1_cfg4_local dist1_cfg5_prefetch1 dist1_cfg5_prefetch2 dist1_cfg5_local; do python - <<PY
import json
try:
    d = json.loads(open('$out/$f.json').read().strip().splitlines()[-1])
    print('$f', round(d['ms_per_step'], 5), round(d['value']), (d.get('roofline_alt') or {}).get('ms_per_step_alt'))
except Exception as e:
    print('$f', 'unreadable', e)
PY
done
tail -14 $out/bench_cfg3_timeline.txt
cat $out/lookup_share_ab.txt
if [ -f scratch/variants/stamps.so ]; then
  TT_LIB_PATH=$PWD/scratch/variants/stamps.so timeout -k 10 200 python scratch/opt_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-600 > $out/optimizer_stamps.txt
fi
timeout -k 10 100 python scratch/bench_gemm.py > $out/gemm_launches.json 2>/dev/null
timeout -k 10 300 python bench_kernels.py > $out/kernels_largeB.jsonl 2>/dev/null
