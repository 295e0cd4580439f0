#!/bin/bash
# leaf postponement threshold, shade / refill thresholds in the round-1 final regime (32-frame launches)
cd $GRAFT_REPO_ROOT
export PB_VARY=1 PB_BATCH=32
for K in 1 2 3 4 6 8 12 16 1; do echo -n "leaf=$K: "; PT_TUNE_LEAF=$K timeout -k 10 120 python tools/pipeline_bench.py 256 2>&1 | grep -v amdgpu.ids | sed 's/tiles 1\/1 batch=32 slots=default: //'; done
for SF in "8 8" "12 8" "16 8" "8 12" "8 16" "12 12" "16 16" "6 6" "4 8"; do set -- $SF; echo -n "shade=$1 fill=$2: "; PT_TUNE_SHADE=$1 PT_TUNE_FILL=$2 timeout -k 10 120 python tools/pipeline_bench.py 256 2>&1 | grep -v amdgpu.ids | sed 's/tiles 1\/1 batch=32 slots=default: //'; done
