#!/bin/bash
cd $GRAFT_REPO_ROOT
for P in 1 2 3; do for F in 16 24 32 48; do
  echo -n "passes=$P flush=$F adopt=0 : "; PT_TUNE_PASSES=$P PT_TUNE_FLUSH=$F timeout -k 10 60 python tools/profile_frame.py 5 | tail -1 | tr "\n" " "; PT_TUNE_PASSES=$P PT_TUNE_FLUSH=$F timeout -k 10 60 python tools/pipeline_bench.py 100
done; done
echo -n "baseline flush=0: "; PT_TUNE_FLUSH=0 timeout -k 10 60 python tools/pipeline_bench.py 100
for P in 1 2; do echo -n "1/8 passes=$P flush=32: "; PB_TILES=8 PT_TUNE_GRIDDIV=4 PT_TUNE_PASSES=$P PT_TUNE_FLUSH=32 timeout -k 10 60 python tools/pipeline_bench.py 200; done
echo -n "1/8 flush=0: "; PB_TILES=8 PT_TUNE_GRIDDIV=4 PT_TUNE_FLUSH=0 timeout -k 10 60 python tools/pipeline_bench.py 200
