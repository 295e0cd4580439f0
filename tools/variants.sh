#!/bin/bash
# Rebuild the library with each set of -D flags given as arguments (one quoted string per variant) and time C2 / C2 solo / C4 with
# each, all in one gpurun session.  usage: tools/variants.sh "-DPT_SHORT_STACK=6 -DPT_MEGA_WAVES_PER_SIMD=6" "-DPT_FETCH_DMA=0" ...
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  echo "== build: $V"
  make -s -B -j8 -C raytracer-public_amd/csrc EXTRA="$V" 2>&1 | grep -E "error" && continue
  bash tools/quick_ab.sh
done
make -s -B -j8 -C raytracer-public_amd/csrc 2>&1 | grep -E "error"
