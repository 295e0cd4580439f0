#!/bin/bash
# chunk size (items per queue claim) under the frame-aligned queue order
cd $GRAFT_REPO_ROOT
export PB_VARY=1
for B in 32 8 1; do for K in 512 128 256 1024 2048 512; do
  echo -n "batch=$B chunk=$K: "; PT_TUNE_CHUNK=$K PB_BATCH=$B timeout -k 10 120 python tools/pipeline_bench.py $((B*8 > 96 ? B*8 : 96)) 2>&1 | grep -v "amdgpu.ids" | sed 's/tiles 1\/[0-9]* batch=[0-9]* slots=default: //' || exit 1
done; done
