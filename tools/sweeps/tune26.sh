#!/bin/bash
# shade threshold (lanes waiting to be shaded before a shade pass runs), 32-frame and single-frame launches
cd $GRAFT_REPO_ROOT
export PB_VARY=1
for B in 32 1; do for S in 8 16 20 24 32 40 16 8; do echo -n "batch=$B shade=$S: "; PB_BATCH=$B PT_TUNE_SHADE=$S timeout -k 10 120 python tools/pipeline_bench.py $((B*8 > 96 ? B*8 : 96)) 2>&1 | grep -v amdgpu.ids | sed 's/tiles 1\/1 batch=[0-9]* slots=default: //'; done; done
