#!/bin/bash
# Same-session A/B of one launch heuristic (PT_TUNE_<KNOB>) on the launches that matter (the set of tools/ab/kvariants.sh), current build.
#   usage (GPU box): tools/ab/tune_ab.sh <out file under gpurun_out> KNOB v1 v2 ...      e.g. tools/ab/tune_ab.sh quad.txt QUAD 0 16 0 16
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; KNOB=$2; shift 2; mkdir -p $(dirname $OUT); : > $OUT
run() { "$@" 2>&1 | tail -1; }
for V in "$@"; do
  echo "== PT_TUNE_$KNOB = $V" >> $OUT
  export PT_TUNE_$KNOB=$V
  { echo -n "batch32   "; PB_BATCH=32 PB_VARY=1 run python3 tools/pipeline_bench.py 128
    echo -n "solo20    "; PB_SOLO=1 PB_BATCH=20 PB_VARY=1 run python3 tools/pipeline_bench.py 80
    echo -n "solo8     "; PB_SOLO=1 PB_BATCH=8 PB_VARY=1 run python3 tools/pipeline_bench.py 48
    echo -n "solo1     "; PB_SOLO=1 PB_BATCH=1 PB_VARY=1 run python3 tools/pipeline_bench.py 40
    echo -n "pipe1     "; PB_BATCH=1 PB_VARY=1 run python3 tools/pipeline_bench.py 40
    echo -n "share8x20 "; PB_SOLO=1 PB_TILES=8 PB_BATCH=20 PB_VARY=1 run python3 tools/pipeline_bench.py 80
    echo -n "share8x5  "; PB_SOLO=1 PB_TILES=8 PB_BATCH=5 PB_VARY=1 run python3 tools/pipeline_bench.py 40
    echo -n "sponza    "; PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 run python3 tools/pipeline_bench.py 24
  } >> $OUT
done
cat $OUT
