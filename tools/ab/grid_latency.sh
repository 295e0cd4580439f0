#!/bin/bash
# How long is ONE wavefront's traversal step?  The instrumented launch (tools/wave_timeline.py) of a lone C2 frame with the grid cut to
# 1/div of the residency (PT_TUNE_GRIDDIV): 6144 / div wavefronts = 6 / div per SIMD.  If a step's time were issue contention between
# the wavefronts of a SIMD it would fall with div; what stays is the dependent latency of one wavefront's own step.
#   usage (GPU box): tools/ab/grid_latency.sh <out file under gpurun_out>
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $(dirname $OUT); : > $OUT
for D in 1 2 3 6 12 24; do
  echo "== grid = residency / $D  ($((6144 / D)) wavefronts)" >> $OUT
  PT_TUNE_GRIDDIV=$D python3 tools/wave_timeline.py 2>&1 | grep -E "^ms|frames in|queue-empty|^end|pre-exhaustion|post-exhaustion|cycles per" >> $OUT
  echo -n "   un-instrumented lone frame: " >> $OUT
  PT_TUNE_GRIDDIV=$D PB_SOLO=1 PB_BATCH=1 PB_VARY=1 python3 tools/pipeline_bench.py 24 2>&1 | tail -1 >> $OUT
done
cat $OUT
