#!/bin/bash
# A/B of the two megakernel generations in one gpurun session (box-to-box variance is ~10 %): C2 in 32-frame launches, C2 one
# render() per frame, C4.  usage: tools/ab_kernels.sh [extra PT_TUNE_* assignments for the generation-2 runs]
cd $GRAFT_REPO_ROOT
for K in 1 2 1 2; do
  echo "== kernel generation $K"
  PT_TUNE_KERNEL=$K PB_BATCH=32 PB_VARY=1 python3 tools/pipeline_bench.py 192 2>&1 | tail -1
  PT_TUNE_KERNEL=$K PB_BATCH=1 PB_VARY=1 python3 tools/pipeline_bench.py 60 2>&1 | tail -1
  PT_TUNE_KERNEL=$K PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 python3 tools/pipeline_bench.py 32 2>&1 | tail -1
done
