#!/bin/bash
# build megakernel variants on the GPU box and time them: tools/variants.sh "<extra flags>" ...
cd $GRAFT_REPO_ROOT/raytracer-public_amd/csrc
for V in "$@"; do
  make -s clean; make -s FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -I../../include $V" 2>&1 | grep -E "error" | head -3
  echo -n "[$V] : "; (cd $GRAFT_REPO_ROOT && PB_BATCH=8 timeout -k 10 60 python tools/pipeline_bench.py 240)
done
make -s clean; make -s 2>&1 | grep -E "error" | head -3
