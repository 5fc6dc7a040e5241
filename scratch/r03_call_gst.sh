#!/bin/bash
TT_LIB_PATH=$PWD/scratch/variants/gstamps.so timeout -k 10 200 python scratch/gemm_stamps_bwd.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
