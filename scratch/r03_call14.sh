#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_tower" 2>&1 | tail -5
bash scratch/r03_ab_lib.sh tower_fwd2,gemm_kernel main pfg8 main pfg8
export TT_FUSED_TOWER=0
bash scratch/r03_ab_lib.sh tower_fwd2,gemm_kernel main
