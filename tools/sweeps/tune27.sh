#!/bin/bash
# deferred triangle tests: on/off and thresholds
cd $GRAFT_REPO_ROOT
export PB_VARY=1
run() { echo -n "batch=$1 defer=$2 leaf=$3 wait=$4: "; PB_BATCH=$1 PT_TUNE_DEFER=$2 PT_TUNE_DEFER_LEAF=$3 PT_TUNE_DEFER_WAIT=$4 timeout -k 10 120 python tools/pipeline_bench.py $(($1*8 > 96 ? $1*8 : 96)) 2>&1 | grep -v amdgpu.ids | sed 's/tiles 1\/1 batch=[0-9]* slots=default: //'; }
run 32 0 24 4
for L in 8 16 24 32 40 48; do for W in 2 4 8 16; do run 32 1 $L $W; done; done
run 32 0 24 4
run 1 0 24 4; run 1 1 24 4; run 1 1 32 8; run 8 0 24 4; run 8 1 24 4
