#!/bin/bash
out=gpurun_out/r03; mkdir -p $out
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $out/gpu_tests10.log 2>&1; rc=$?; echo "rc=$rc" >> $out/gpu_tests10.log
tail -5 $out/gpu_tests10.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/gpu_tests10.log | head -30; exit 1; }
echo "== plain cfg3"; python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
echo "== sharded world=1 no collectives cfg3"; TT_FORCE_DIST=1 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
for c in cfg3 cfg4 cfg5; do
  echo "== sharded one rank, forced RCCL calls, $c"
  TT_FORCE_DIST=1 TT_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config $c 2>/dev/null | grep '^{' > $out/dist1_$c.json
  grep -o '"ms_per_step": [0-9.]*' $out/dist1_$c.json
done
