#!/bin/bash
# frame slots for sharded batches
cd $GRAFT_REPO_ROOT
for T in 8 2; do for B in 8 32; do for S in 2 3 4; do
  PB_RING=1 PB_TILES=$T PB_BATCH=$B PT_TUNE_SLOTS=$S timeout -k 10 120 python tools/pipeline_bench.py $((B*12)) || exit 1
done; done; done
for B in 2 4; do for S in 2 3; do
  PB_RING=1 PB_BATCH=$B PT_TUNE_SLOTS=$S timeout -k 10 120 python tools/pipeline_bench.py $((B*40)) || exit 1
done; done
