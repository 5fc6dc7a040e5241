#!/bin/bash
TT_LIB_PATH=$PWD/scratch/variants/stamps.so timeout -k 10 200 python scratch/opt_stamps.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500
