#!/bin/bash
bash scratch/r03_ab_opt.sh main pf2 main pf2 2>&1 | grep -v "latest WGs\|dense blocks" | tail -30
