#!/bin/bash
# XCD-aware queue (8 ranges, one per XCD, with hopping) vs the single queue: throughput, then L2 hit rate / HBM traffic by PMC
cd $GRAFT_REPO_ROOT
export PB_VARY=1 PB_RING=1
for B in 32 8 1; do for X in 0 1 0 1; do
  echo -n "xcd=$X "; PT_TUNE_XCD=$X PB_BATCH=$B timeout -k 10 120 python tools/pipeline_bench.py $((B*8 > 64 ? B*8 : 64)) 2>&1 | grep -v "amdgpu.ids\|^ring" || exit 1
done; done
echo -n "xcd=1 tiles 1/8: "; PT_TUNE_XCD=1 PB_TILES=8 PB_BATCH=32 timeout -k 10 120 python tools/pipeline_bench.py 384 2>&1 | grep -v "amdgpu.ids\|^ring"
echo -n "xcd=0 tiles 1/8: "; PT_TUNE_XCD=0 PB_TILES=8 PB_BATCH=32 timeout -k 10 120 python tools/pipeline_bench.py 384 2>&1 | grep -v "amdgpu.ids\|^ring"
