#!/bin/bash
# integer rows of the batch transposition: rows = frames per launch (aligned frames) vs frames x segments
cd $GRAFT_REPO_ROOT
export PB_VARY=1 PB_RING=1
run() { echo -n "tiles=1/$1 batch=$2 rows=$3: "; PB_TILES=$1 PT_TUNE_ROWS=$3 PB_BATCH=$2 timeout -k 10 120 python tools/pipeline_bench.py $(($2*8 > 96 ? $2*8 : 96)) 2>&1 | grep -v "amdgpu.ids\|^ring" | sed 's/tiles 1\/[0-9]* batch=[0-9]* slots=default: //' || exit 1; }
for R in 64 32 16; do run 1 32 $R; done
for R in 64 24 8 12 48; do run 1 24 $R; done
for R in 64 12 4 6 24; do run 1 12 $R; done
for R in 64 8 16 512 1024; do run 1 8 $R; done
for R in 64 4 8 512 1024; do run 1 4 $R; done
for R in 64 3 96 384; do run 1 3 $R; done
for R in 64 256 512; do run 1 2 $R; done
for R in 64 128 192; do run 1 1 $R; done
for R in 64 32 256 1024 4096; do run 8 32 $R; done
for R in 64 8 256 1024; do run 8 8 $R; done
for R in 64 128 256; do run 8 1 $R; done
