#!/bin/bash
# Same-session A/B of the working tree against the library under tools/ab/raytracer_base (tools/ab/README), alternating, N rounds:
# 32-frame launches, 20-frame launches with a sync (the driver's command), a lone frame, one render() per frame, C4.
#   usage (GPU box): tools/ab/base_ab.sh <out file under gpurun_out> [rounds]
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; N=${2:-2}; mkdir -p $(dirname $OUT); : > $OUT
run() { "$@" 2>&1 | tail -1; }
shapes() {
  echo -n "batch32   "; PB_BATCH=32 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 192
  echo -n "solo20    "; PB_SOLO=1 PB_BATCH=20 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 80
  echo -n "solo1     "; PB_SOLO=1 PB_BATCH=1 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 40
  echo -n "pipe1     "; PB_BATCH=1 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 40
  echo -n "sponza    "; PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 run timeout -k 5 100 python3 tools/pipeline_bench.py 32
}
for i in $(seq $N); do
  echo "== base (tools/ab/raytracer_base)" >> $OUT; PB_BASE=1 shapes >> $OUT
  echo "== tree" >> $OUT; shapes >> $OUT
done
cat $OUT
