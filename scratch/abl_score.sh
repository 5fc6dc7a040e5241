#!/bin/bash
# build score.hip variants with extra -D flags: args "name:-DFLAG=... -DFLAG2=..." -> scratch/sc/lib<name>.so
mkdir -p /root/repo/scratch/sc
cd /root/repo/two_tower_amazon_recommender_amd/csrc
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $f -c score.hip -o /tmp/score_$n.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/sc/lib$n.so build/capi_common.o build/fill.o build/gather.o build/route.o build/encode.o build/sort.o build/sparse.o build/gemm.o /tmp/score_$n.o build/dense_update.o ) &
done
wait
ls /root/repo/scratch/sc/
