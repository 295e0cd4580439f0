#!/bin/bash
cd $GRAFT_REPO_ROOT
for Q in 12 16 24; do for S in 8 10 12; do for T in 8 4 1; do
  echo -n "hwq=$Q "; GPU_MAX_HW_QUEUES=$Q PB_TILES=$T PT_TUNE_SLOTS=$S timeout -k 10 60 python tools/pipeline_bench.py 200
done; done; done
