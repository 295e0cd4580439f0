#!/bin/bash
cd $GRAFT_REPO_ROOT
for T in 8 4 2; do for B in 1 2 4 8; do for S in 2 3 4; do
  PB_TILES=$T PB_BATCH=$B PT_TUNE_SLOTS=$S timeout -k 10 60 python tools/pipeline_bench.py 400
done; done; done
for B in 1 2 3 4; do for S in 2 3; do PB_TILES=1 PB_BATCH=$B PT_TUNE_SLOTS=$S timeout -k 10 60 python tools/pipeline_bench.py 200; done; done
