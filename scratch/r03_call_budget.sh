#!/bin/bash
mkdir -p gpurun_out/r03m
timeout -k 10 300 python scratch/budget_cfg4_n8.py 2> gpurun_out/r03m/budget.err | tee gpurun_out/r03m/budget_cfg4_n8.json
tail -5 gpurun_out/r03m/budget.err | grep -v amdgpu
