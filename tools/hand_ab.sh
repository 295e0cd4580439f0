#!/bin/bash
# Same-session A/B of the path hand-over knobs on launches that END (lone frame, 20-frame launch, 1/8 shares), pipelined ones and C4.
#   usage (GPU box): tools/hand_ab.sh <out file under gpurun_out> "HANDOVER:HANDAFTER" ...     e.g. tools/hand_ab.sh h.txt 0:0 1:0 1:128
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift; mkdir -p $(dirname $OUT); : > $OUT
run() { "$@" 2>&1 | tail -1; }
for V in "$@"; do
  export PT_TUNE_HANDOVER=${V%%:*} PT_TUNE_HANDAFTER=${V##*:}
  echo "== HANDOVER = $PT_TUNE_HANDOVER HANDAFTER = $PT_TUNE_HANDAFTER" >> $OUT
  { echo -n "solo1     "; PB_SOLO=1 PB_BATCH=1 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 40
    echo -n "solo20    "; PB_SOLO=1 PB_BATCH=20 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 80
    echo -n "share8x5  "; PB_SOLO=1 PB_TILES=8 PB_BATCH=5 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 40
    echo -n "pipe1     "; PB_BATCH=1 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 40
    echo -n "batch32   "; PB_BATCH=32 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 128
    echo -n "sponza    "; PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 24
  } >> $OUT
done
cat $OUT
