#!/bin/bash
for kb in 0 40 76 100; do
  export TT_TOWER_LDS_KB=$kb
  echo "##### TT_TOWER_LDS_KB=$kb"
  bash scratch/r03_ab_lib.sh tower_fwd2 main
done
