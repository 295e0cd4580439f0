#!/bin/bash
# launches that END (a sync after every launch): 20-frame launches, 8-frame launches and solo frames -- what a short run or a single frame costs incl. its drain
cd $GRAFT_REPO_ROOT
PB_SOLO=1 PB_BATCH=20 PB_VARY=1 python3 tools/pipeline_bench.py 80 2>&1 | tail -1 | sed 's/^/solo20 /'
PB_SOLO=1 PB_BATCH=8 PB_VARY=1 python3 tools/pipeline_bench.py 48 2>&1 | tail -1 | sed 's/^/solo8 /'
PB_SOLO=1 PB_BATCH=1 PB_VARY=1 python3 tools/pipeline_bench.py 40 2>&1 | tail -1 | sed 's/^/solo1 /'
PB_BATCH=1 PB_VARY=1 python3 tools/pipeline_bench.py 40 2>&1 | tail -1 | sed 's/^/pipe1 /'
