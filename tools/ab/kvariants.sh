#!/bin/bash
# Same-session A/B of megakernel build variants (-D flags) on the launches that matter: 32-frame launches (throughput), 20-frame
# launches with a sync (the driver's command), a lone frame, one render() per frame without waits, a 1/8 tile share in 20-frame
# launches, and C4.   usage (GPU box): tools/ab/kvariants.sh <out file under gpurun_out> "<flags of variant 1>" "<flags of variant 2>" ...
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; shift; mkdir -p $(dirname $OUT); : > $OUT
run() { "$@" 2>&1 | tail -1; }
for V in "$@"; do
  echo "== variant: ${V:-<default>}" >> $OUT
  make -s -B -j8 -C raytracer-public_amd/csrc EXTRA="$V" 2>&1 | grep -E "error" >> $OUT && continue
  # the build under test is bit-exact before it is timed (a wrong kernel's time means nothing)
  if [ -n "$KV_FULL" ]; then timeout -k 10 400 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -1 >> $OUT       # KV_FULL=1: the whole GPU suite per variant
  else timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -q -x -k "path_mode_bit_exact or sponza_class or stack_overflow or quad_mode" 2>&1 | tail -1 >> $OUT; fi
  { echo -n "batch32   "; PB_BATCH=32 PB_VARY=1 run python3 tools/pipeline_bench.py 128
    echo -n "solo20    "; PB_SOLO=1 PB_BATCH=20 PB_VARY=1 run python3 tools/pipeline_bench.py 80
    echo -n "solo1     "; PB_SOLO=1 PB_BATCH=1 PB_VARY=1 run python3 tools/pipeline_bench.py 40
    echo -n "pipe1     "; PB_BATCH=1 PB_VARY=1 run python3 tools/pipeline_bench.py 40
    echo -n "share8x20 "; PB_SOLO=1 PB_TILES=8 PB_BATCH=20 PB_VARY=1 run python3 tools/pipeline_bench.py 80
    echo -n "sponza    "; PF_SCENE=sponza PB_BATCH=8 PB_VARY=1 run python3 tools/pipeline_bench.py 24
  } >> $OUT
done
make -s -B -j8 -C raytracer-public_amd/csrc 2>&1 | grep -E "error"
cat $OUT
