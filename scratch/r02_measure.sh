#!/bin/bash
# Round-2 measurement set -> gpurun_out/r02/ (copied into profiles/ afterwards)
out=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "== bench cfg3 (default flags)"; timeout -k 10 300 python bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err || echo FAILED
for c in cfg1 cfg2; do
  echo "== bench $c"; timeout -k 10 200 python bench.py --config $c --no-cpu-baseline > $out/bench_$c.json 2>/dev/null || echo FAILED
  echo "== bench $c --graph"; timeout -k 10 200 python bench.py --config $c --graph --no-cpu-baseline > $out/bench_${c}_graph.json 2>/dev/null || echo FAILED
done
echo "== bench cfg3 --graph"; timeout -k 10 200 python bench.py --graph --no-cpu-baseline > $out/bench_cfg3_graph.json 2>/dev/null || echo FAILED
echo "== bench cfg4 (un-sharded, one GPU)"; timeout -k 10 300 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_cfg4.json 2>/dev/null || echo FAILED
echo "== bench cfg5 (un-sharded, one GPU)"; timeout -k 10 400 python bench.py --config cfg5 --steps 20 --warmup 5 --ids Z > $out/bench_cfg5.json 2>/dev/null || echo FAILED
echo "== dist leg, one rank, forced RCCL calls"
for c in cfg3 cfg4 cfg5; do
  TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config $c 2>/dev/null | grep '^{' > $out/dist1_$c.json || echo FAILED
done
echo "== kernel microbenches"
timeout -k 10 200 python scratch/bench_k2.py > $out/k2_batch.jsonl 2>/dev/null
timeout -k 10 200 python scratch/bench_sort.py > $out/sort_plan.jsonl 2>/dev/null
timeout -k 10 300 python bench_kernels.py > $out/kernels_largeB.jsonl 2>/dev/null
timeout -k 10 100 python scratch/bench_gemm.py > $out/gemm_launches.json 2>/dev/null
timeout -k 10 200 python scratch/bench_score.py > $out/score_f32_vs_bf16x3.jsonl 2>/dev/null
echo "== rocprof kernel trace"
bash scratch/prof.sh r02b --steps 200 --warmup 20 > /dev/null 2>&1
cp gpurun_out/prof_r02b/trace_kernel_stats.csv $out/bench_cfg3_kernel_stats.csv; cp gpurun_out/prof_r02b/bench.json $out/bench_cfg3_under_rocprof.json
python scratch/timeline.py gpurun_out/prof_r02b/trace_kernel_trace.csv > $out/bench_cfg3_timeline.txt 2>&1
echo "== PMC passes"
bash scratch/prof_pmc.sh r02 --steps 40 --warmup 10 > /dev/null 2>&1
python scratch/pmc_summary.py gpurun_out/pmc_r02 $out/pmc_cfg3_sgd.json
ls -la $out
