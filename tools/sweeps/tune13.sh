#!/bin/bash
# frame slots vs batch size: throughput and per-launch duration (overlapped launches stretch each other)
cd $GRAFT_REPO_ROOT
for B in 8 16 32; do for S in 1 2 3; do
  PB_RING=1 PB_BATCH=$B PT_TUNE_SLOTS=$S timeout -k 10 120 python tools/pipeline_bench.py $((B*10)) || exit 1
done; done
