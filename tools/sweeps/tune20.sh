#!/bin/bash
# rows of the batch transposition (2^r) x XCD-aware queue on/off, for several frames-per-launch
cd $GRAFT_REPO_ROOT
export PB_VARY=1 PB_RING=1
for B in 1 2 4 8 16 32; do
  for X in 0 1; do for R in 4 5 6 7 8; do
    echo -n "batch=$B xcd=$X rows_log2=$R: "; PT_TUNE_XCD=$X PT_TUNE_ROWS=$R PB_BATCH=$B timeout -k 10 120 python tools/pipeline_bench.py $((B*8 > 96 ? B*8 : 96)) 2>&1 | grep -v "amdgpu.ids\|^ring" | sed 's/tiles 1\/1 batch=[0-9]* slots=default: //' || exit 1
  done; done
done
