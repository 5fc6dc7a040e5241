#!/bin/bash
# forced split counts at the N = 8 per-GPU slab of cfg4 (2048 x 16384 x 128): pass 1 (ns_q; model: 32) x dc pass (ns_cs; model: 4)
cd $GRAFT_REPO_ROOT
for q in 32 16 24 48 64; do for cs in 4 2 3 6 8; do
  echo -n "ns_q $q ns_cs $cs  "
  TT_NSPLIT_Q=$q TT_NSPLIT_CS=$cs timeout -k 10 120 python scratch/r04_slab.py 2048x16384x128 --iters 30 2>/dev/null | tail -1 | cut -c1-300
done; done
