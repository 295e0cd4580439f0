#!/bin/bash
# rows of the batch transposition (2^r) with the XCD-aware queue, 32-frame launches; and for solo frames
cd $GRAFT_REPO_ROOT
export PB_VARY=1 PB_RING=1
for R in 6 3 4 5 7 8 6; do echo -n "rows_log2=$R "; PT_TUNE_ROWS=$R PB_BATCH=32 timeout -k 10 120 python tools/pipeline_bench.py 256 2>&1 | grep -v "amdgpu.ids\|^ring" || exit 1; done
for R in 6 4 5 7 8 6; do echo -n "rows_log2=$R "; PT_TUNE_ROWS=$R PB_BATCH=1 timeout -k 10 120 python tools/pipeline_bench.py 100 2>&1 | grep -v "amdgpu.ids\|^ring" || exit 1; done
