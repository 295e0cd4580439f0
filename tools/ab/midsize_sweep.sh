#!/bin/bash
# Launch heuristics (rows of the batch transposition, XCD-aware queue) for MID-SIZE launches: 2.5 .. 10 frames of work per launch -- what the sharded runs of bench.py
# and 8-frame batches submit.  One session; every line = tools/pipeline_bench.py with the knobs in its label.
#   usage (GPU box): tools/ab/midsize_sweep.sh <out file under gpurun_out>
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $(dirname $OUT); : > $OUT
run() { "$@" 2>&1 | tail -1; }
# tiles batch frames : (share count, frames per launch, frames to run)
for SHAPE in "1 8 48" "2 10 60" "2 5 40" "4 10 80" "4 5 40" "8 5 40" "8 10 80" "8 20 80"; do
  set -- $SHAPE; T=$1; B=$2; F=$3
  for ROWS in auto 1 128; do
    for XCD in auto 0 1; do
      if [ $ROWS = auto ]; then unset PT_TUNE_ROWS; else export PT_TUNE_ROWS=$((ROWS * B)); fi
      if [ $XCD = auto ]; then unset PT_TUNE_XCD; else export PT_TUNE_XCD=$XCD; fi
      echo -n "tiles 1/$T batch $B rows/frame=$ROWS xcd=$XCD: " >> $OUT
      PB_SOLO=1 PB_TILES=$T PB_BATCH=$B PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py $F >> $OUT
    done
  done
done
cat $OUT
