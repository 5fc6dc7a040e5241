#!/bin/bash
# Round-3 measurement set, part B: kernel microbenches and the same-box A/Bs that get a committed artefact
out=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "== kernel microbenches"
timeout -k 10 200 python scratch/bench_k2.py > $out/k2_batch.jsonl 2>/dev/null
timeout -k 10 200 python scratch/bench_sort.py > $out/sort_plan.jsonl 2>/dev/null
timeout -k 10 300 python bench_kernels.py > $out/kernels_largeB.jsonl 2>/dev/null
timeout -k 10 100 python scratch/bench_gemm.py > $out/gemm_launches.json 2>/dev/null
timeout -k 10 400 python scratch/bench_score.py 8192x128 16384x128 8192x256 32768x256 > $out/score_f32_vs_bf16x3.jsonl 2>/dev/null
echo "== A/B: piece sums written through (sc1) vs the r02 release fence per piece (variant fence)"
for v in main fence; do
  if [ "$v" = main ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$PWD/scratch/variants/$v.so; fi
  timeout -k 10 300 python bench_kernels.py --only table 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); d['variant']='$v'
    if 'sparse_apply' in d['kernel']: print(json.dumps(d))" >> $out/sparse_apply_ab.jsonl
done
unset TT_LIB_PATH
echo "== A/B: dc pass (BWD_S) with 8-wave (HEAD) vs 4-wave workgroups"
for v in main bwds4 main bwds4; do
  if [ "$v" = main ]; then unset TT_LIB_PATH; else export TT_LIB_PATH=$PWD/scratch/variants/$v.so; fi
  bash scratch/prof.sh ab_$v --steps 200 --warmup 20 > /dev/null 2>&1
  echo "-- $v" >> $out/ab_dc_pass_waves.txt
  python scratch/kstats.py gpurun_out/prof_ab_$v/trace_kernel_stats.csv "score_kernel<128, 5" "score_kernel<128, 4" >> $out/ab_dc_pass_waves.txt
  python -c "
import json; print('ms_per_step', json.loads(open('gpurun_out/prof_ab_$v/bench.json').read().strip().splitlines()[-1])['ms_per_step'])" >> $out/ab_dc_pass_waves.txt
  rm -f gpurun_out/prof_ab_$v/trace_kernel_trace.csv
done
unset TT_LIB_PATH
echo "== the lambda loop form (variant lam = -DTT_LOOP_LAMBDA=1) through the parity / determinism tests"
TT_LIB_PATH=$PWD/scratch/variants/lam.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "retrieval_baseline_configs or retrieval_is_deterministic or rank_pass_is_deterministic or odd_and_short" 2>&1 | tail -3 > $out/lam_tests.txt
cat $out/lam_tests.txt
echo "== optimizer stamps (HEAD)"
TT_LIB_PATH=$PWD/scratch/variants/stamps.so timeout -k 10 200 python scratch/opt_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-600 > $out/optimizer_stamps.txt
TT_LIB_PATH=$PWD/scratch/variants/stamps.so timeout -k 10 200 python scratch/tower_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-600 > $out/tower_stamps.txt || true
cat $out/ab_dc_pass_waves.txt; cat $out/sparse_apply_ab.jsonl | cut -c1-220; cat $out/optimizer_stamps.txt | head -3; cat $out/tower_stamps.txt | head -5
cat $out/kernels_largeB.jsonl | cut -c1-200
cat $out/score_f32_vs_bf16x3.jsonl | cut -c1-700
