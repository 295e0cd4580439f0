#!/bin/bash
# launch chaining (next launch starts when the previous queue is dry) vs free overlap, by frame slots
cd $GRAFT_REPO_ROOT
export PB_VARY=1 PB_RING=1
run() { echo -n "tiles=1/$1 batch=$2 chain=$3 slots=$4: "; PB_TILES=$1 PT_TUNE_CHAIN=$3 PT_TUNE_SLOTS=$4 PB_BATCH=$2 timeout -k 10 120 python tools/pipeline_bench.py $(($2*8 > 96 ? $2*8 : 96)) 2>&1 | grep -v "amdgpu.ids" | tr '\n' ' ' | sed 's/tiles 1\/[0-9]* batch=[0-9]* slots=[0-9]*: //'; echo; }
for S in 2 3; do for C in 1 0; do run 1 32 $C $S; done; done
for S in 2 3 4; do for C in 1 0; do run 1 8 $C $S; done; done
for S in 2 3 4 6; do for C in 1 0; do run 1 1 $C $S; done; done
for S in 2 3 4; do for C in 1 0; do run 8 32 $C $S; done; done
for S in 3 4 8; do for C in 1 0; do run 8 8 $C $S; done; done
for S in 4 8 12; do for C in 1 0; do run 8 1 $C $S; done; done
