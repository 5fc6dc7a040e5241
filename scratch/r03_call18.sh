#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_tower" 2>&1 | tail -5
TT_LIB_PATH=$PWD/scratch/variants/tstamps.so python scratch/tower_stamps.py 2>&1 | grep "workgroups\|max:"
bash scratch/r03_ab_lib.sh tower_fwd2,gemm_kernel main main
