#!/bin/bash
# build timing-only ablation variants of the bf16x3 scorer (bits: 1 no staging split, 2 GEMM1 1/8, 4 GEMM2 1/4, 8 no coef mid)
cd /root/repo/two_tower_amazon_recommender_amd/csrc
for v in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DTT_BX3_ABL=$v -c score.hip -o /tmp/score_abl$v.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/bx3/libabl$v.so build/capi_common.o build/fill.o build/gather.o build/route.o build/encode.o build/sort.o build/sparse.o build/gemm.o /tmp/score_abl$v.o build/dense_update.o ) &
done
wait
ls -la /root/repo/scratch/bx3/*.so
