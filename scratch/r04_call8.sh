#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c8; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_trainer.py -x -q -m gpu -k "match_oracle or composite or asymmetric" > $O/pytest_trainer.txt 2>&1 || { tail -30 $O/pytest_trainer.txt; exit 1; }
tail -2 $O/pytest_trainer.txt
TT_LIB_PATH=$R/scratch/variants/gstamps.so timeout -k 10 300 python scratch/gemm_stamps_bwd.py 2>&1 | grep -v amdgpu.ids > $O/gemm_stamps_bwd.txt
cat $O/gemm_stamps_bwd.txt
python bench.py --config ref --steps 400 --warmup 40 --no-cpu-baseline > $O/bench_ref.json 2> $O/bench_ref.err
python -c "
import json; d=json.load(open('$O/bench_ref.json')); print('ref ms/step', d['ms_per_step'], d['value']/1e6)"
