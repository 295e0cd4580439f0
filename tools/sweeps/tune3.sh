#!/bin/bash
cd $GRAFT_REPO_ROOT
for Q in 4 8 16; do for S in 6 8 12 16; do
  echo -n "hwq=$Q "; GPU_MAX_HW_QUEUES=$Q PB_TILES=8 PT_TUNE_SLOTS=$S PT_TUNE_GRIDDIV=4 timeout -k 10 60 python tools/pipeline_bench.py 200
done; done
for Q in 8 16; do for S in 6 8 12; do
  echo -n "hwq=$Q "; GPU_MAX_HW_QUEUES=$Q PB_TILES=1 PT_TUNE_SLOTS=$S timeout -k 10 60 python tools/pipeline_bench.py 100
done; done
