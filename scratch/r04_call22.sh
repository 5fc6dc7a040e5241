#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c22; mkdir -p $O
cd $R
for i in 1 2; do
python scratch/r04_slab.py 8192x8192x128 2048x16384x128 4096x4096x64 > $O/head_$i.jsonl 2>/dev/null
TT_LIB_PATH=$R/scratch/variants/tpb4.so python scratch/r04_slab.py 8192x8192x128 2048x16384x128 4096x4096x64 > $O/tpb4_$i.jsonl 2>/dev/null
done
for f in head_1 tpb4_1 head_2 tpb4_2; do echo $f; python -c "
import sys,json
for l in open('$O/$f.jsonl'):
    d=json.loads(l); print(' ', d['nq'],d['nc'],d['dim'],'p1',d['pass1_us'],'p2', d['pass2_us'],'tot', d['kernels_us'])"; done
TT_LIB_PATH=$R/scratch/variants/tpb4.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "retrieval" 2>&1 | tail -2
