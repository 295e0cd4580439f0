#!/bin/bash
# XCD-aware queue variants: 0 single queue, 1 column ranges of the 64-row transposition, 2 contiguous batch ranges with 2^r rows
cd $GRAFT_REPO_ROOT
export PB_VARY=1 PB_RING=1
for B in 32 8 1; do
  for CFG in "0 0" "1 0" "2 0" "2 2" "2 3" "2 4" "2 6" "0 0" "1 0"; do
    set -- $CFG
    echo -n "mode=$1 rows_log2=$2 "; PT_TUNE_XCD=$1 PT_TUNE_XCDROWS=$2 PB_BATCH=$B timeout -k 10 120 python tools/pipeline_bench.py $((B*8 > 64 ? B*8 : 64)) 2>&1 | grep -v "amdgpu.ids\|^ring" || exit 1
  done
done
