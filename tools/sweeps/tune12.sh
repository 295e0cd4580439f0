#!/bin/bash
# larger batches for sharded frames (1/8, 1/4, 1/2 shares) and whole frames
cd $GRAFT_REPO_ROOT
for T in 8 4 2 1; do for B in 8 16 32; do
  PB_TILES=$T PB_BATCH=$B timeout -k 10 120 python tools/pipeline_bench.py $((B*12)) || exit 1
done; done
