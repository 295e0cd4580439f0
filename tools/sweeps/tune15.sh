#!/bin/bash
# solo frames (one slot, no batching): donation of sparse wavefronts' paths to continuation passes
cd $GRAFT_REPO_ROOT
export PT_TUNE_SLOTS=1 PB_RING=1
echo -n "base: "; timeout -k 10 120 python tools/pipeline_bench.py 60 || exit 1
for F in 8 16 24 32 48; do for P in 1 2 3; do
  echo -n "flush=$F passes=$P: "; PT_TUNE_FLUSH=$F PT_TUNE_PASSES=$P timeout -k 10 120 python tools/pipeline_bench.py 60 || exit 1
done; done
