#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c21; mkdir -p $O
cd $R
for i in 1 2; do
python scratch/r04_slab.py 8192x8192x128 > $O/head_$i.jsonl 2>/dev/null
TT_LIB_PATH=$R/scratch/variants/abl_nostage.so python scratch/r04_slab.py 8192x8192x128 > $O/abl_$i.jsonl 2>/dev/null
done
cat $O/head_1.jsonl $O/abl_1.jsonl $O/head_2.jsonl $O/abl_2.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['pass1_us'], d['pass2_us'], d['kernels_us'])"
