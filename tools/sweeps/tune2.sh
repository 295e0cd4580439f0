#!/bin/bash
cd $GRAFT_REPO_ROOT
export PT_TUNE_SLOTS=6
for C in 128 256 512 1024 2048; do echo -n "chunk=$C "; PT_TUNE_CHUNK=$C timeout -k 10 60 python tools/pipeline_bench.py 60; done
for SH in 4 8 16 24 32; do for FI in 4 8 16 32; do echo -n "shade=$SH fill=$FI "; PT_TUNE_SHADE=$SH PT_TUNE_FILL=$FI timeout -k 10 60 python tools/pipeline_bench.py 60; done; done
