#!/bin/bash
# more points for the rows rule: non power-of-two batches, small batches with few / many rows, tile shares
cd $GRAFT_REPO_ROOT
export PB_VARY=1 PB_RING=1
run() { echo -n "tiles=1/$1 batch=$2 xcd=$3 rows_log2=$4: "; PB_TILES=$1 PT_TUNE_XCD=$3 PT_TUNE_ROWS=$4 PB_BATCH=$2 timeout -k 10 120 python tools/pipeline_bench.py $(($2*8 > 96 ? $2*8 : 96)) 2>&1 | grep -v "amdgpu.ids\|^ring" | sed 's/tiles 1\/[0-9]* batch=[0-9]* slots=default: //' || exit 1; }
for R in 3 8 9; do run 1 8 1 $R; done
for R in 2 9; do run 1 4 0 $R; done
for R in 9; do run 1 2 0 $R; done
for R in 3 4 5 6 8; do run 1 24 1 $R; done
for R in 3 4 6 8; do run 1 12 1 $R; done
for X in 0 1; do for R in 4 5 6 8; do run 8 32 $X $R; done; done
for R in 4 6 7 8; do run 8 8 0 $R; done
for R in 4 5 6 8; do run 2 32 1 $R; done
