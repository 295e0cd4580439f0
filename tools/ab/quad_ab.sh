#!/bin/bash
# Same-session A/B of the drain's quad mode (round 4): the library of the previous commit (tools/ab/README), and the working tree with
#   PT_TUNE_QUAD=0   one ray per lane throughout
#   defaults         paths re-seated one per quad once a wavefront has nothing left to start and <= 16 paths
# (the run recorded in profiles/r04_q1_quad_vs_consolidation_ab.txt also had round 3's drain consolidation in the tree: PT_TUNE_CONSOLIDATE)
# on the launches that matter.  usage (GPU box): tools/ab/quad_ab.sh <out file under gpurun_out>
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $(dirname $OUT); : > $OUT
run() { "$@" 2>&1 | tail -1; }
suite() {
  echo -n "batch32   "; PB_BATCH=32 PB_VARY=1 run python3 tools/pipeline_bench.py 128
  echo -n "solo20    "; PB_SOLO=1 PB_BATCH=20 PB_VARY=1 run python3 tools/pipeline_bench.py 80
  echo -n "solo8     "; PB_SOLO=1 PB_BATCH=8 PB_VARY=1 run python3 tools/pipeline_bench.py 64
  echo -n "solo1     "; PB_SOLO=1 PB_BATCH=1 PB_VARY=1 run python3 tools/pipeline_bench.py 40
  echo -n "pipe1     "; PB_BATCH=1 PB_VARY=1 run python3 tools/pipeline_bench.py 40
  echo -n "share8x20 "; PB_SOLO=1 PB_TILES=8 PB_BATCH=20 PB_VARY=1 run python3 tools/pipeline_bench.py 80
  echo -n "share8x5  "; PB_SOLO=1 PB_TILES=8 PB_BATCH=5 PB_VARY=1 run python3 tools/pipeline_bench.py 40
  echo -n "sponza    "; PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 run python3 tools/pipeline_bench.py 24
  echo -n "sponza1   "; PF_SCENE=sponza PB_SOLO=1 PB_BATCH=1 PB_VARY=1 run python3 tools/pipeline_bench.py 8
}
if [ -f tools/ab/raytracer_base/libmi355pt.so ]; then echo "== base (previous commit)" >> $OUT; PB_BASE=1 suite >> $OUT; fi
echo "== tree: QUAD=0 (one ray per lane throughout)" >> $OUT; PT_TUNE_QUAD=0 suite >> $OUT
echo "== tree: defaults (quad drain)" >> $OUT; suite >> $OUT
for Q in $QUAD_LIVES; do echo "== tree: QUAD=$Q" >> $OUT; PT_TUNE_QUAD=$Q suite >> $OUT; done
cat $OUT
