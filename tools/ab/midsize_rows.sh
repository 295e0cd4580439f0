#!/bin/bash
# Where does "one row per frame" start to beat "128 segments per frame"?  Whole frames and half shares, 3 .. 8 frames of work per launch, isolated launches.
#   usage (GPU box): tools/ab/midsize_rows.sh <out file under gpurun_out>
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $(dirname $OUT); : > $OUT
run() { "$@" 2>&1 | tail -1; }
for SHAPE in "1 3 36" "1 4 40" "1 5 40" "1 6 48" "1 7 56" "2 6 48" "2 8 48" "2 12 72" "2 14 84" "4 16 96" "4 20 80" "4 24 96"; do
  set -- $SHAPE; T=$1; B=$2; F=$3
  for ROWS in 1 128; do
    export PT_TUNE_ROWS=$((ROWS * B)) PT_TUNE_XCD=0
    echo -n "tiles 1/$T batch $B rows/frame=$ROWS xcd=0: " >> $OUT
    PB_SOLO=1 PB_TILES=$T PB_BATCH=$B PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py $F >> $OUT
  done
done
cat $OUT
