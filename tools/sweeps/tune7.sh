#!/bin/bash
cd $GRAFT_REPO_ROOT
export PB_BATCH=4
for S in 2 3 4; do echo -n "slots=$S: "; PT_TUNE_SLOTS=$S timeout -k 10 60 python tools/pipeline_bench.py 240; done
for SH in 4 8 16 24; do for FI in 4 8 16 24; do echo -n "shade=$SH fill=$FI: "; PT_TUNE_SHADE=$SH PT_TUNE_FILL=$FI timeout -k 10 60 python tools/pipeline_bench.py 240; done; done
for C in 128 256 512 1024; do echo -n "chunk=$C: "; PT_TUNE_CHUNK=$C timeout -k 10 60 python tools/pipeline_bench.py 240; done
for B in 4 6 8; do echo -n "batch=$B: "; PB_BATCH=$B timeout -k 10 60 python tools/pipeline_bench.py 240; done
