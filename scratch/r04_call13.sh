#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c13; mkdir -p $O
cd $R
python scratch/r04_dw_check.py > $O/dw_check_direct.txt 2>&1
TT_LIB_PATH=$R/scratch/variants/dwtile.so python scratch/r04_dw_check.py > $O/dw_check_tile.txt 2>&1
echo "== direct"; grep -v amdgpu $O/dw_check_direct.txt; echo "== tile"; grep -v amdgpu $O/dw_check_tile.txt
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/gputests.txt 2>&1; tail -5 $O/gputests.txt
