#!/bin/bash
timeout -k 10 200 python scratch/bwd_pieces.py 2>&1 | grep -v amdgpu.ids
