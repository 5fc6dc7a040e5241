#!/bin/bash
# build exact-f32 scorer variants: args "W:G" = waves per workgroup : target workgroups per pass -> scratch/sw/libw<W>g<G>.so
cd /root/repo/two_tower_amazon_recommender_amd/csrc
for v in "$@"; do
  w=${v%%:*}; g=${v##*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DTT_SCORE_WAVES=$w -DTT_SCORE_WGS=$g -c score.hip -o /tmp/score_w${w}g${g}.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/sw/libw${w}g${g}.so build/capi_common.o build/fill.o build/gather.o build/route.o build/encode.o build/sort.o build/sparse.o build/gemm.o /tmp/score_w${w}g${g}.o build/dense_update.o ) &
done
wait
ls -la /root/repo/scratch/sw/*.so
