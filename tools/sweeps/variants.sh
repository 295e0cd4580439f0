#!/bin/bash
# build megakernel variants on the GPU box and time them: tools/variants.sh "<extra flags>" ...
cd $GRAFT_REPO_ROOT/raytracer-public_amd/csrc
for V in "$@"; do
  make -s clean; make -s FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -I../../include $V" 2>&1 | grep -E "error" | head -3
  for B in ${VAR_BATCHES:-8}; do echo -n "[$V] batch=$B: "; (cd $GRAFT_REPO_ROOT && PB_VARY=1 PB_BATCH=$B timeout -k 10 90 python tools/pipeline_bench.py $((B*8 > 96 ? B*8 : 96)) 2>&1 | grep -v amdgpu.ids); done
done
make -s clean; make -s 2>&1 | grep -E "error" | head -3
